"""TEST INFRASTRUCTURE ONLY (oracle/): numpy ray caster of the segmentation camera (csrc/sdx_camera.hip): the same boxes (brick
bounding boxes with id = brick index + 1, statics and robot boxes with id 0), the same pinhole model (position, target, world z up,
square image, horizontal field of view), nearest slab hit per pixel.  PARITY UNPINNED against Isaac Gym's renderer (closed; renders
the studded meshes): this oracle only checks that the kernel does what DESIGN.md says it does."""
import numpy as np

F = np.float32


def quat_apply(q, v):
    u = q[..., :3]
    t = 2.0 * np.cross(u, v)
    return v + q[..., 3:4] * t + np.cross(u, t)


def quat_mul(a, b):
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], axis=-1)


def scene_boxes(desc, root_env, rb_env):
    """(centres [B,3], quats [B,4], halves [B,3], ids [B]) of one env from sdx_scene_desc + its ROOT [142,13] / RB [165,13] rows"""
    bt = np.array(list(desc.brick_type))
    bc = np.array([list(r) for r in desc.brick_center], F)
    bh = np.array([list(r) for r in desc.brick_half], F)
    r = root_env[9:141].astype(F)
    q = r[:, 3:7]
    c = [r[:, 0:3] + quat_apply(q, bc[bt])]
    qs, hs, ids = [q], [bh[bt]], [np.arange(1, 133)]
    ns = desc.n_static
    if ns:
        c.append(np.array([list(desc.static_center[s]) for s in range(ns)], F))
        qs.append(np.tile(np.array([[0, 0, 0, 1]], F), (ns, 1)))
        hs.append(np.array([list(desc.static_half[s]) for s in range(ns)], F))
        ids.append(np.zeros(ns, int))
    nr = desc.n_rbox
    if nr:
        link = np.array(list(desc.rbox_link))[:nr]
        ql = rb_env[link, 3:7].astype(F)
        c.append(rb_env[link, 0:3].astype(F) + quat_apply(ql, np.array([list(desc.rbox_center[k]) for k in range(nr)], F)))
        qs.append(quat_mul(ql, np.array([list(desc.rbox_quat[k]) for k in range(nr)], F)))
        hs.append(np.array([list(desc.rbox_half[k]) for k in range(nr)], F))
        ids.append(np.zeros(nr, int))
    return np.concatenate(c).astype(F), np.concatenate(qs).astype(F), np.concatenate(hs).astype(F), np.concatenate(ids)


def render(desc, root_env, rb_env, W=128, H=128):
    c, q, h, ids = scene_boxes(desc, root_env, rb_env)
    cam, tgt = np.array(list(desc.seg_cam_pos), F), np.array(list(desc.seg_cam_target), F)
    f = tgt - cam
    f = f / np.linalg.norm(f)
    r = np.cross(f, np.array([0, 0, 1], F))
    r = r / np.linalg.norm(r)
    u = np.cross(r, f)
    th = np.tan(0.5 * np.deg2rad(desc.seg_cam_hfov_deg))
    cols, rows = np.meshgrid(np.arange(W), np.arange(H))
    px = ((2.0 * (cols + 0.5) / W - 1.0) * th).astype(F)
    py = ((1.0 - 2.0 * (rows + 0.5) / H) * th).astype(F)
    d = (f[None, None] + px[..., None] * r[None, None] + py[..., None] * u[None, None]).astype(F)        # [H, W, 3]
    eye = np.eye(3, dtype=F)
    best = np.full((H, W), 3.0e38, F)
    img = np.zeros((H, W), np.int16)
    for b in range(c.shape[0]):
        M = np.stack([quat_apply(q[b], eye[a]) for a in range(3)])                                       # rows = box axes in world coordinates
        o = M @ (cam - c[b])
        dl = d @ M.T
        tmin = np.zeros((H, W), F)
        tmax = np.full((H, W), 3.0e38, F)
        hit = np.ones((H, W), bool)
        for a in range(3):
            da = dl[..., a]
            par = np.abs(da) < 1e-12
            hit &= ~(par & (abs(o[a]) > h[b, a]))
            inv = 1.0 / np.where(par, 1.0, da)
            t0, t1 = (-h[b, a] - o[a]) * inv, (h[b, a] - o[a]) * inv
            lo, hi = np.minimum(t0, t1), np.maximum(t0, t1)
            tmin = np.where(par, tmin, np.maximum(tmin, lo))
            tmax = np.where(par, tmax, np.minimum(tmax, hi))
        hit &= (tmin <= tmax) & (tmin < best)
        best = np.where(hit, tmin, best)
        img = np.where(hit, ids[b], img).astype(np.int16)
    return img


def pixel_stats(img, target_id):
    rows, cols = np.nonzero(img == target_id)
    n = rows.size
    return n, (int(rows.astype(np.float32).mean()) if n else 0), (int(cols.astype(np.float32).mean()) if n else 0)
