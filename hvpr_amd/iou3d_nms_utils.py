"""Python face of the reference's absent native module pcdet/ops/iou3d_nms/iou3d_nms_utils.py, same function names and
argument meaning ([upstream] signatures), backed by the HIP kernels.  Boxes: [x,y,z,dx,dy,dz,heading]."""
import torch

from . import kernels

_WS = {}


def _ws(n, device):
    key = (int(n), device)
    if key not in _WS:
        _WS.clear()
        _WS[key] = kernels.PostWorkspace(1, max(int(n), 1), max(int(n), 1), device)
    return _WS[key]


def boxes_iou_bev(boxes_a, boxes_b):
    return kernels.boxes_pairwise(boxes_a.float(), boxes_b.float(), 1)


def boxes_overlap_bev(boxes_a, boxes_b):
    return kernels.boxes_pairwise(boxes_a.float(), boxes_b.float(), 0)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    return kernels.boxes_pairwise(boxes_a.float(), boxes_b.float(), 2)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Returns (LongTensor keep indices into `boxes`, None); order: descending score, ascending index on ties."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.long, device=boxes.device), None
    if n > 8192:
        raise ValueError("nms_gpu: at most 8192 candidates (NMS_PRE_MAXSIZE is 4096 in the hvpr configs)")
    take = n if pre_maxsize is None else min(int(pre_maxsize), n)
    ws = _ws(n, boxes.device)
    order, _, cnt = kernels.score_topk(scores.float().reshape(1, n).contiguous(), None, take, ws, want_scores=False)
    b = boxes.float().contiguous()
    keep, kc = kernels.nms_bev(b, order[0].contiguous(), cnt, take, float(thresh), take, ws.nms)
    return keep[: int(kc.item())].long(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned BEV NMS ([upstream] nms_normal_gpu: IoU of the boxes with their headings ignored).  Runs the rotated kernel on
    heading-zeroed copies: the overlap of two axis-aligned rectangles through the polygon routine equals the closed form up to
    fp32 rounding.  Returns (LongTensor keep indices, None) like nms_gpu."""
    b = boxes[:, :7].float().clone()
    b[:, 6] = 0.0
    return nms_gpu(b, scores, thresh, **kwargs)
