"""Device-side counterpart of the reference's ``gymnasium_planar_robotics/utils/geometry_2D_utils.py`` (GEO:line): same three
functions, same argument meaning, batched f64 torch tensors on an MI355X in, bool / f64 tensors out. The arithmetic (incl. the
float32 quaternion normalisation of ``rotations_utils.unit_vector``, ROT:447, and the 1e-7 tolerances) is done by the HIP
kernels of libgpr_hip.so; nothing here computes on the CPU."""
from __future__ import annotations

import numpy as np
import torch

from .. import _native


def _dev(a, device) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        t = a.to(dtype=torch.float64)
        if device is not None:
            t = t.to(device)
    else:
        t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=device or 'cuda')
    if t.device.type != 'cuda':
        raise RuntimeError('geometry_2D_utils runs on MI355X only: pass CUDA/ROCm tensors (there is no CPU path)')
    return t.contiguous()


def check_line_segments_intersect(p1, p2, q1, q2, device=None) -> torch.Tensor:
    """GEO:9-69. p1, p2, q1, q2: (num_checks, 2). Returns bool (num_checks,)."""
    p1 = _dev(p1, device); device = p1.device
    p2, q1, q2 = _dev(p2, device), _dev(q1, device), _dev(q2, device)
    n = p1.shape[0]
    assert p1.shape == (n, 2) and p2.shape == (n, 2) and q1.shape == (n, 2) and q2.shape == (n, 2)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _native.check(_native.load().gpr_check_line_segments_intersect(n, p1.data_ptr(), p2.data_ptr(), q1.data_ptr(), q2.data_ptr(),
                                                                       out.data_ptr(), _native.current_stream_ptr(device)))
    return out.bool()


def get_2D_rect_vertices(qpos, size, device=None) -> torch.Tensor:
    """GEO:72-104. qpos (num_rectangles, 7), size (num_rectangles, 2) half sizes. Returns (num_rectangles, 2, 4)."""
    qpos = _dev(qpos, device); device = qpos.device
    size = _dev(size, device)
    n = qpos.shape[0]
    assert qpos.shape == (n, 7) and size.shape == (n, 2)
    out = torch.empty((n, 2, 4), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        _native.check(_native.load().gpr_get_2D_rect_vertices(n, qpos.data_ptr(), size.data_ptr(), out.data_ptr(),
                                                              _native.current_stream_ptr(device)))
    return out


def check_rectangles_intersect(qpos_r1, qpos_r2, size_r1, size_r2, device=None) -> torch.Tensor:
    """GEO:107-138 (edge crossing only: a rectangle strictly inside another does not intersect it). Returns bool (num_checks,)."""
    a = _dev(qpos_r1, device); device = a.device
    b, c, d = _dev(qpos_r2, device), _dev(size_r1, device), _dev(size_r2, device)
    n = a.shape[0]
    assert a.shape == (n, 7) and b.shape == (n, 7) and c.shape == (n, 2) and d.shape == (n, 2)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _native.check(_native.load().gpr_check_rectangles_intersect(n, a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), out.data_ptr(),
                                                                    _native.current_stream_ptr(device)))
    return out.bool()
