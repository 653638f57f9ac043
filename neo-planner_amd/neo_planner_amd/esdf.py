"""
Device-backed maps with the reference's map protocol (map_server/esdf.py:7-82):
`occupancy_map_cb`, `get_edt_dis`, `get_edt_grad`, `has_collision`, `is_occuiped` and the
attributes `esdf_map`, `esdf_grad_x`, `esdf_grad_y`, `occupancy_2d`, `map_resolution`,
`map_origin.{x,y}`, `map_width`, `map_height`.

The distance transform and the gradient run on the MI355X (neo_esdf_build_2d); the arrays
are copied back once so that the attribute protocol of the reference keeps working.
"""
import ctypes
import types

import numpy as np

from . import _lib

SAFE_DIS = 0.5      # esdf.py:4


class ESDF:
    """drop-in for map_server/esdf.py:ESDF"""

    def __init__(self, ctx=None):
        self.ctx = ctx if ctx is not None else _lib.default_context()
        self.scene_id = self.ctx.new_scene_id()
        self.version = 0            # bumped on every map update: planners re-snapshot when it changes

    def occupancy_map_cb(self, map):
        """esdf.py:11-33.  `map` is a nav_msgs/OccupancyGrid (or anything shaped like one)."""
        self.map_resolution = map.info.resolution
        self.map_width = int(map.info.width)
        self.map_height = int(map.info.height)
        self.map_origin = map.info.origin.position
        occ = np.ascontiguousarray(np.asarray(map.data, dtype=np.int8).reshape(self.map_height, self.map_width))
        self.occupancy_2d = (occ == 100).astype(np.int64)
        n = (self.map_height, self.map_width)
        dist, gx, gy = np.empty(n), np.empty(n), np.empty(n)
        c = self.ctx
        c.check(c.lib.neo_esdf_build_2d(c.h, self.scene_id, _lib.ptr(occ), self.map_width, self.map_height,
                                        float(self.map_resolution), float(self.map_origin.x),
                                        float(self.map_origin.y), _lib.ptr(dist), _lib.ptr(gx), _lib.ptr(gy)))
        self.esdf_map, self.esdf_grad_x, self.esdf_grad_y = dist, gx, gy
        self.version += 1

    @classmethod
    def from_arrays(cls, esdf_map, grad_x, grad_y, resolution, origin_xy, ctx=None):
        """upload precomputed arrays (esdf.py:29-33 results) without rebuilding them"""
        self = cls(ctx)
        self.esdf_map = _lib.as_f64(esdf_map)
        self.esdf_grad_x = _lib.as_f64(grad_x)
        self.esdf_grad_y = _lib.as_f64(grad_y)
        self.map_height, self.map_width = self.esdf_map.shape
        self.map_resolution = resolution
        self.map_origin = types.SimpleNamespace(x=float(origin_xy[0]), y=float(origin_xy[1]))
        c = self.ctx
        c.check(c.lib.neo_esdf_upload_2d(c.h, self.scene_id, _lib.ptr(self.esdf_map), _lib.ptr(self.esdf_grad_x),
                                         _lib.ptr(self.esdf_grad_y), self.map_width, self.map_height,
                                         float(resolution), self.map_origin.x, self.map_origin.y))
        self.version += 1
        return self

    def _query(self, pos):
        p = np.ascontiguousarray(np.asarray(pos, dtype=np.float64).reshape(1, -1)[:, :2])
        d = np.empty(1)
        g = np.empty((1, 2))
        c = self.ctx
        c.check(c.lib.neo_esdf_query(c.h, self.scene_id, 1, _lib.ptr(p), _lib.ptr(d), _lib.ptr(g)))
        return d, g

    def query(self, pts):
        """batched lookups: pts (n, 2) -> dist (n,), grad (n, 2)"""
        p = np.ascontiguousarray(np.asarray(pts, dtype=np.float64)[:, :2])
        d = np.empty(len(p))
        g = np.empty((len(p), 2))
        c = self.ctx
        c.check(c.lib.neo_esdf_query(c.h, self.scene_id, len(p), _lib.ptr(p), _lib.ptr(d), _lib.ptr(g)))
        return d, g

    def get_edt_dis(self, pos):                     # esdf.py:53-67
        d, _ = self._query(pos)
        return 10000 if d[0] == 10000.0 else d[0]

    def get_edt_grad(self, pos):                    # esdf.py:69-82
        d, g = self._query(pos)
        if d[0] == 10000.0 and g[0, 0] == 0.0 and g[0, 1] == 0.0:
            return [0, 0]
        return [g[0, 0], g[0, 1]]

    def has_collision(self, pos):                   # esdf.py:50-51
        return self.get_edt_dis(pos) < SAFE_DIS

    def is_occuiped(self, pos):                     # esdf.py:35-48 (sic)
        row = int((pos[1] - self.map_origin.y) / self.map_resolution)
        col = int((pos[0] - self.map_origin.x) / self.map_resolution)
        if row < 0 or row >= self.map_height or col < 0 or col >= self.map_width:
            return False
        return self.occupancy_2d[row, col]


class ESDF3D:
    """3-D distance field for the trilinear mode (north-star configs; no reference counterpart).
    dist[z, y, x] at voxel centres origin + (i + 0.5) * resolution."""

    def __init__(self, dist, resolution, origin_xyz, store="f32", layout="linear", ctx=None):
        self.ctx = ctx if ctx is not None else _lib.default_context()
        self.scene_id = self.ctx.new_scene_id()
        self.version = 1
        self.resolution = float(resolution)
        self.origin = np.asarray(origin_xyz, dtype=np.float64)
        if dist is None:
            return
        src_dev = False
        try:
            import torch
            if isinstance(dist, torch.Tensor):
                src_dev = dist.is_cuda
                self.shape = tuple(dist.shape)
                src_dtype = {torch.float64: _lib.NEO_F64, torch.float32: _lib.NEO_F32}[dist.dtype]
                dist = dist.contiguous()
                pointer = ctypes.c_void_p(dist.data_ptr())
        except ImportError:
            pass
        if not src_dev and not hasattr(dist, "data_ptr"):
            dist = np.ascontiguousarray(dist)
            if dist.dtype not in (np.float64, np.float32):
                dist = dist.astype(np.float32)
            self.shape = dist.shape
            src_dtype = _lib.NEO_F64 if dist.dtype == np.float64 else _lib.NEO_F32
            pointer = _lib.ptr(dist)
        nz, ny, nx = self.shape
        org = (ctypes.c_double * 3)(*self.origin)
        c = self.ctx
        c.check(c.lib.neo_esdf_upload_3d(c.h, self.scene_id, pointer, src_dtype, int(src_dev), nx, ny, nz,
                                         self.resolution, ctypes.cast(org, ctypes.c_void_p),
                                         {"f32": _lib.NEO_F32, "f16": _lib.NEO_F16}[store],
                                         _lib.LAYOUTS[layout]))

    @classmethod
    def from_occupancy(cls, occ, resolution, origin_xyz, store="f32", layout="linear", ctx=None, want_dist=False):
        """occupancy [z, y, x] (non-zero = occupied; NumPy uint8 or a torch CUDA uint8 tensor) -> exact EDT
        on the GPU (3-D counterpart of occupancy_map_cb).  `want_dist` copies the float32 field back
        into `self.dist`."""
        self = cls(None, resolution, origin_xyz, ctx=ctx)
        on_dev = False
        try:
            import torch
            if isinstance(occ, torch.Tensor):
                occ = occ.contiguous()
                assert occ.dtype == torch.uint8
                on_dev = occ.is_cuda
                self.shape = tuple(occ.shape)
                pointer = ctypes.c_void_p(occ.data_ptr())
                if not on_dev:
                    occ = occ.numpy()
        except ImportError:
            pass
        if not on_dev:
            occ = np.ascontiguousarray(occ, dtype=np.uint8)
            self.shape = occ.shape
            pointer = _lib.ptr(occ)
        nz, ny, nx = self.shape
        self.dist = np.empty(self.shape, dtype=np.float32) if want_dist else None
        org = (ctypes.c_double * 3)(*self.origin)
        c = self.ctx
        c.check(c.lib.neo_esdf_build_3d(c.h, self.scene_id, pointer, int(on_dev), nx, ny, nz, self.resolution,
                                        ctypes.cast(org, ctypes.c_void_p),
                                        {"f32": _lib.NEO_F32, "f16": _lib.NEO_F16}[store],
                                        _lib.LAYOUTS[layout],
                                        _lib.ptr(self.dist)))
        return self

    def query(self, pts):
        p = np.ascontiguousarray(np.asarray(pts, dtype=np.float64).reshape(-1, 3))
        d = np.empty(len(p))
        g = np.empty((len(p), 3))
        c = self.ctx
        c.check(c.lib.neo_esdf_query(c.h, self.scene_id, len(p), _lib.ptr(p), _lib.ptr(d), _lib.ptr(g)))
        return d, g

    def get_edt_dis(self, pos):
        return self.query(pos)[0][0]

    def get_edt_grad(self, pos):
        return list(self.query(pos)[1][0])
