"""yaml -> attribute dict, with the reference's `_BASE_CONFIG_` include and recursive merge semantics
(pcdet/config.py:51-85).  easydict is not a dependency: AttrDict is the small subset the models use."""
import os

import yaml


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, AttrDict):
        return AttrDict(v)
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    return v


def _merge(config, new, base_dir):
    if "_BASE_CONFIG_" in new:
        path = new["_BASE_CONFIG_"]
        if not os.path.isabs(path):
            for root in (os.getcwd(), base_dir, os.path.dirname(base_dir)):
                if os.path.exists(os.path.join(root, path)):
                    path = os.path.join(root, path)
                    break
        with open(path) as f:
            config.update(AttrDict(yaml.safe_load(f)))
    for k, v in new.items():
        if not isinstance(v, dict):
            config[k] = v
            continue
        if k not in config:
            config[k] = AttrDict()
        _merge(config[k], v, base_dir)
    return config


def cfg_from_yaml_file(cfg_file, config=None):
    config = AttrDict() if config is None else config
    with open(cfg_file) as f:
        new = yaml.safe_load(f)
    return _merge(config, new, os.path.dirname(os.path.dirname(os.path.abspath(cfg_file))))


CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfgs")


def hvpr_car_cfg():
    return cfg_from_yaml_file(os.path.join(CFG_DIR, "kitti_models", "hvpr_car.yaml"))


def hvpr_3class_cfg():
    """BASELINE.json configs[3]: Car / Pedestrian / Cyclist, 6 anchors per location."""
    return cfg_from_yaml_file(os.path.join(CFG_DIR, "kitti_models", "hvpr_3class.yaml"))
