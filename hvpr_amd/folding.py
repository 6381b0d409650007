"""Eval-mode BatchNorm folding shared by the modules: y = x*s + t with s = gamma/sqrt(var+eps), t = beta - mean*s."""
import torch


def bn_scale_shift(bn):
    s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    return s.float(), (bn.bias.detach() - bn.running_mean.detach() * s).float()


class FoldCache:
    """Folded weights are rebuilt when the module returns to eval mode, after load_state_dict, on device moves, or on
    an explicit invalidate(); mutating parameters by hand while in eval mode needs invalidate()."""

    def __init__(self):
        self.value = None
        self.device = None

    def invalidate(self):
        self.value = None

    def get(self, device, build):
        if self.value is None or self.device != device:
            with torch.no_grad():
                self.value = build()
            self.device = device
        return self.value
