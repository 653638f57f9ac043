"""
Synthetic benchmark inputs: random-forest maps and replan requests.

The forest follows the distribution of the reference's world generator
(src/simulator/scripts/generator_config.yaml:1-16, generate_worlds.py:100-146):
axis-aligned box pillars, footprint U[0.5,1.5] m, height U[3,6] m, centre
x in U[3,27], y in U[-5,5], re-drawn until 1.8 m clear of every earlier box.
Domain 30 x 30 x 30 m at 0.1 m leaf size (README.md:140), x in [0,30],
y in [-15,15], z in [0,30].

Everything here is host-side NumPy: it builds *inputs* (occupancy grids,
distance fields, start/goal states, initial guesses).  No planner arithmetic.
"""
import numpy as np

RES = 0.1
DOMAIN_ORIGIN = (0.0, -15.0, 0.0)
DOMAIN_CELLS = 300
PROJECT_Z_RANGE = (1.8, 10.0)       # launch/map_server_onboard.launch:31-32 (occupancy_min_z/max_z)


def forest_boxes(scene_seed, count=None):
    """list of (cx, cy, sx, sy, sz) pillars for scene `scene_seed`."""
    rng = np.random.default_rng(1000 + scene_seed)
    if count is None:
        count = int(rng.choice([10, 15, 20]))
    boxes = []
    for _ in range(count):
        sx, sy = rng.uniform(0.5, 1.5, 2)
        sz = rng.uniform(3.0, 6.0)
        for _attempt in range(10000):
            cx = rng.uniform(3.0, 27.0)
            cy = rng.uniform(-5.0, 5.0)
            ok = True
            for (bx, by, bsx, bsy, _bsz) in boxes:
                if abs(cx - bx) < (sx + bsx) / 2 + 1.8 and abs(cy - by) < (sy + bsy) / 2 + 1.8:
                    ok = False
                    break
            if ok:
                boxes.append((cx, cy, sx, sy, sz))
                break
    return boxes


def canopy_boxes(scene_seed, count=80):
    """floating boxes for the 3-D scenes (cx, cy, cz, sx, sy, sz): clutter above and between the pillars so that the
    distance field varies along z everywhere a 3-D request flies (the reference's worlds are pillars only,
    generate_worlds.py:100-146, because its planner is planar; the north-star scenes are 300^3 volumes).
    Centres x in U[2,28], y in U[-13,13], z in U[3,27]; edges U[0.4,2.0] m; no clearance rule."""
    rng = np.random.default_rng(3000 + scene_seed)
    out = []
    for _ in range(count):
        sx, sy, sz = rng.uniform(0.4, 2.0, 3)
        out.append((rng.uniform(2.0, 28.0), rng.uniform(-13.0, 13.0), rng.uniform(3.0, 27.0), sx, sy, sz))
    return out


def _axis_range(lo, hi, origin, n, res=RES):
    i0 = int(np.floor((lo - origin) / res))
    i1 = int(np.ceil((hi - origin) / res))
    return max(i0, 0), min(i1, n)


def occupancy_2d(scene_seed, n=DOMAIN_CELLS, count=None, unknown_frac=0.0, res=RES):
    """projected 2-D occupancy grid, row-major [row=y, col=x], int8 values
    100 (occupied) / 0 (free) / -1 (unknown), as an OccupancyGrid.data array
    (map_server/esdf.py:16-26).  Pillars taller than PROJECT_Z_RANGE[0] project."""
    occ = np.zeros((n, n), dtype=np.int8)
    for (cx, cy, sx, sy, sz) in forest_boxes(scene_seed, count):
        if sz < PROJECT_Z_RANGE[0]:
            continue
        x0, x1 = _axis_range(cx - sx / 2, cx + sx / 2, DOMAIN_ORIGIN[0], n, res)
        y0, y1 = _axis_range(cy - sy / 2, cy + sy / 2, DOMAIN_ORIGIN[1], n, res)
        occ[y0:y1, x0:x1] = 100
    if unknown_frac > 0:
        rng = np.random.default_rng(5000 + scene_seed)
        mask = (rng.random(occ.shape) < unknown_frac) & (occ == 0)
        occ[mask] = -1
    return occ


def occupancy_3d(scene_seed, n=DOMAIN_CELLS, count=None, res=RES, canopy=0):
    """[z, y, x] uint8 occupancy (1 = occupied) of the forest plus the ground slab z < res; `canopy` > 0 adds that
    many floating boxes (canopy_boxes)."""
    occ = np.zeros((n, n, n), dtype=np.uint8)
    for (cx, cy, cz, sx, sy, sz) in (canopy_boxes(scene_seed, canopy) if canopy else []):
        x0, x1 = _axis_range(cx - sx / 2, cx + sx / 2, DOMAIN_ORIGIN[0], n, res)
        y0, y1 = _axis_range(cy - sy / 2, cy + sy / 2, DOMAIN_ORIGIN[1], n, res)
        z0, z1 = _axis_range(cz - sz / 2, cz + sz / 2, DOMAIN_ORIGIN[2], n, res)
        occ[z0:z1, y0:y1, x0:x1] = 1
    for (cx, cy, sx, sy, sz) in forest_boxes(scene_seed, count):
        x0, x1 = _axis_range(cx - sx / 2, cx + sx / 2, DOMAIN_ORIGIN[0], n, res)
        y0, y1 = _axis_range(cy - sy / 2, cy + sy / 2, DOMAIN_ORIGIN[1], n, res)
        z0, z1 = _axis_range(0.0, sz, DOMAIN_ORIGIN[2], n, res)
        occ[z0:z1, y0:y1, x0:x1] = 1
    occ[0, :, :] = 1
    return occ


def esdf_3d(scene_seed, n=DOMAIN_CELLS, count=None, dtype=np.float32, res=RES, canopy=0):
    """exact Euclidean distance (metres) to the nearest occupied voxel, [z, y, x]."""
    from scipy import ndimage
    occ = occupancy_3d(scene_seed, n, count, res, canopy)
    return (ndimage.distance_transform_edt(1 - occ) * res).astype(dtype)


VOLUME = dict(z_range=(1.0, 25.0), y_range=(-12.0, 12.0), pitch=0.25)   # replan_requests(**VOLUME): requests that fill the box


def replan_requests(scene_seed, B, n_wpts, D=2, init_T=2.5, z_plane=2.0,
                    length_range=(10.0, 28.0), jitter=0.5, z_range=None, y_range=(-4.0, 4.0), pitch=0.0):
    """B replan requests for one scene:
    head position uniform in x in [0,3], y in [-4,4]; head velocity N(0,0.3);
    tail = head + L * dir with L in U[length_range], heading within +-20 deg of +x
    (clipped to the domain), tail velocity 0; `n_wpts` straight-line waypoints
    with N(0, jitter) lateral noise; durations init_T with first/last x1.5
    (expert_planner.py:96-99).
    D = 3: by default every request sits in the plane z = z_plane (the reference flies at a fixed height).  With
    `z_range` the start height is U[z_range], the path climbs or descends at a pitch angle in U[-pitch, pitch] rad
    (goal height clipped to z_range), the head velocity gets a vertical component and the waypoint noise acts on
    both directions normal to the path: genuinely three-dimensional requests (bench.py's cfg2/cfg4/cfg5).

    returns head[B,3,D], tail[B,3,D], int_wpts[B,D,n_wpts], ts[B,n_wpts+1]  (float64)
    """
    rng = np.random.default_rng(2000 + scene_seed)
    M = n_wpts + 1
    head = np.zeros((B, 3, D))
    tail = np.zeros((B, 3, D))
    head[:, 0, 0] = rng.uniform(0.2, 3.0, B)
    head[:, 0, 1] = rng.uniform(y_range[0], y_range[1], B)
    head[:, 1, :2] = rng.normal(0.0, 0.3, (B, 2))
    L = rng.uniform(length_range[0], length_range[1], B)
    ang = rng.uniform(-0.35, 0.35, B)
    tail[:, 0, 0] = np.clip(head[:, 0, 0] + L * np.cos(ang), 0.5, 29.5)
    tail[:, 0, 1] = np.clip(head[:, 0, 1] + L * np.sin(ang), -14.5, 14.5)
    if D == 3 and z_range is not None:
        head[:, 0, 2] = rng.uniform(z_range[0], z_range[1], B)
        tail[:, 0, 2] = np.clip(head[:, 0, 2] + L * np.tan(rng.uniform(-pitch, pitch, B)), z_range[0], z_range[1])
        head[:, 1, 2] = rng.normal(0.0, 0.3, B)
    elif D == 3:
        head[:, 0, 2] = z_plane
        tail[:, 0, 2] = z_plane
    step = (tail[:, 0, :] - head[:, 0, :]) / (n_wpts + 1)                 # [B, D]
    k = np.arange(1, n_wpts + 1)[None, None, :]                            # [1,1,n]
    wpts = head[:, 0, :, None] + step[:, :, None] * k                      # [B, D, n]
    along = step[:, :2] / np.linalg.norm(step[:, :2], axis=1, keepdims=True)
    lateral = np.stack([-along[:, 1], along[:, 0]], axis=1)                # [B, 2]
    noise = rng.normal(0.0, jitter, (B, n_wpts))
    wpts[:, :2, :] += lateral[:, :, None] * noise[:, None, :]
    if D == 3 and z_range is not None:
        # second normal direction: (step x lateral), mostly vertical
        s3 = step / np.linalg.norm(step, axis=1, keepdims=True)
        lat3 = np.concatenate([lateral, np.zeros((B, 1))], axis=1)
        up = np.cross(s3, lat3)
        wpts += up[:, :, None] * rng.normal(0.0, jitter, (B, n_wpts))[:, None, :]
        wpts[:, 2, :] = np.clip(wpts[:, 2, :], 0.3, 29.7)
    ts = np.full((B, M), init_T)
    ts[:, 0] *= 1.5
    ts[:, -1] *= 1.5
    return head, tail, wpts, ts


class OccupancyGridMsg:
    """duck-typed nav_msgs/OccupancyGrid: .data, .info.resolution/.width/.height/
    .origin.position.{x,y} -- what ESDF.occupancy_map_cb reads (map_server/esdf.py:16-20)."""

    class _NS:
        pass

    def __init__(self, occ2d, resolution=RES, origin_xy=(DOMAIN_ORIGIN[0], DOMAIN_ORIGIN[1])):
        occ2d = np.asarray(occ2d)
        self.data = occ2d.reshape(-1).tolist()
        self.info = self._NS()
        self.info.resolution = resolution
        self.info.height, self.info.width = occ2d.shape
        self.info.origin = self._NS()
        self.info.origin.position = self._NS()
        self.info.origin.position.x = origin_xy[0]
        self.info.origin.position.y = origin_xy[1]
