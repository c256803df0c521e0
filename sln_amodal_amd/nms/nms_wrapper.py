"""Reference surface: nms/nms_wrapper.py:14-17."""
from .pth_nms import pth_nms


def nms(dets, thresh):
    """Greedy NMS.  dets: Tensor[N,5] (y1,x1,y2,x2,score); returns kept indices."""
    return pth_nms(dets, thresh)
