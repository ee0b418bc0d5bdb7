#!/usr/bin/env python3
"""Model compiler: reference asset files -> compact scene-constant table (JSON).

Runs ONLY in the build container (it reads /root/reference/assets, which never
travels to the GPU box).  The output, seqdex_amd/scene_data/grasp_sim_scene.json,
is *derived data* (numbers measured from the URDF / STL / OBJ files), not a copy
of any reference source file.  It is the SURVEY.md §8(a) row A0 input for every
kernel and for the C oracle.

What it derives (reference call sites it replaces in brackets):
  * the 24-body / 23-DOF kinematic tree of franka_panda_allegro.urdf after
    collapse_fixed_joints [GS:540-558], joint frames, axes, limits;
  * per-link mass / COM / inertia: URDF <inertial> where present, otherwise
    density 1000 x convex-hull volume of the collision mesh with the inertia of
    the mesh's bounding box (our documented convention, SURVEY.md §7 hard parts);
  * per-link collision boxes: URDF <box> shapes verbatim; the fingertip and palm meshes as SLAB COMPOUNDS of their convex
    hull (hull_slabs below); the arm / mount / camera meshes, which never touch a brick, as their link-frame bounding box;
  * PD drive gains / effort / velocity limits [GS:580-590];
  * the 8 brick types [assets/urdf/blender/urdf/*.urdf]: mass = 567 x hull volume, centre of mass and inertia of the convex
    hull (what PhysX derives for a single-hull shape, GS:717-731), the bounding box, a slab compound of the hull ("sub" boxes:
    the collision shape of a free brick) and the HOLLOW compound of the mesh itself - four walls, roof, upper part - which is the
    shape of the brick being inserted where the reference runs V-HACD on it (IS:698-709);
  * the base plates 4x4x{1,2,4}_real as stud compounds: body + 16 studs of two boxes each (shaft, chamfered tip) - what the
    V-HACD hulls of GS:810-838 / IS:740-767 resolve;
  * static scene boxes: table, 5 bin walls [GS:629-685], the 60-brick fixed floor merged into one slab [GS:748-808] (its
    studs are not resolved: 180 of them; a hull-shaped brick cannot sink between them);
  * default poses / constants used by the task [GS:249-311, 887-889].
"""
import json
import math
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

REF = "/root/reference/assets/urdf"
ROBOT_URDF = os.path.join(REF, "franka_description/robots/franka_panda_allegro.urdf")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "seqdex_amd",
                   "scene_data", "grasp_sim_scene.json")


# ----------------------------------------------------------------------------- math
def rpy_to_mat(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx  # URDF fixed-axis roll, pitch, yaw


def mat_to_quat_xyzw(m):
    t = np.trace(m)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        w, x, y, z = 0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        w, x, y, z = (m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        w, x, y, z = (m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        w, x, y, z = (m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s
    q = np.array([x, y, z, w])
    return (q / np.linalg.norm(q)).tolist()


def origin_tf(elem):
    """(R, t) of an <origin> element (identity when missing)."""
    if elem is None:
        return np.eye(3), np.zeros(3)
    xyz = [float(v) for v in elem.get("xyz", "0 0 0").split()]
    rpy = [float(v) for v in elem.get("rpy", "0 0 0").split()]
    return rpy_to_mat(*rpy), np.array(xyz)


# --------------------------------------------------------------------------- meshes
def load_obj(path):
    v = []
    with open(path, "r", errors="ignore") as f:
        for line in f:
            if line.startswith("v "):
                v.append([float(x) for x in line.split()[1:4]])
    return np.array(v)


def load_stl(path):
    with open(path, "rb") as f:
        data = f.read()
    ntri = struct.unpack("<I", data[80:84])[0]
    if 84 + 50 * ntri == len(data):  # binary
        arr = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                            count=ntri, offset=84)
        return arr["v"].reshape(-1, 3).astype(np.float64)
    v = []
    for line in data.decode("ascii", errors="ignore").splitlines():
        s = line.strip()
        if s.startswith("vertex"):
            v.append([float(x) for x in s.split()[1:4]])
    return np.array(v)


def resolve_mesh(fname, urdf_dir):
    if fname.startswith("package://franka_description/"):
        return os.path.join(REF, "franka_description", fname[len("package://franka_description/"):])
    return os.path.normpath(os.path.join(urdf_dir, fname))


def load_mesh(path):
    return load_obj(path) if path.lower().endswith(".obj") else load_stl(path)


def box_inertia(m, full):
    x, y, z = full
    return np.diag([m / 12 * (y * y + z * z), m / 12 * (x * x + z * z), m / 12 * (x * x + y * y)])


# ------------------------------------------------------------------ convex hulls: mass properties, slab compounds
def hull_mass_props(v):
    """volume, centroid and inertia tensor about the centroid (unit density) of the convex hull of the points v"""
    h = ConvexHull(v)
    c0 = v[h.vertices].mean(0)
    vol, cen, cov = 0.0, np.zeros(3), np.zeros((3, 3))
    for s in h.simplices:
        P = v[s] - c0
        w = abs(np.dot(P[0], np.cross(P[1], P[2]))) / 6.0          # tetrahedron (c0, a, b, c)
        vol += w
        cen += w * P.sum(0) / 4.0
        S = P.sum(0)
        cov += w / 20.0 * (np.outer(S, S) + P.T @ P)
    cen /= vol
    cov -= vol * np.outer(cen, cen)
    return vol, cen + c0, np.trace(cov) * np.eye(3) - cov


def _hull_sections(v, zs):
    """bounding rectangle (lo, hi) of the convex hull's cross-section at every height of zs"""
    tri = v[ConvexHull(v).simplices]
    out = []
    for z in zs:
        pts = []
        for i in range(3):
            p, q = tri[:, i], tri[:, (i + 1) % 3]
            m = ((p[:, 2] - z) * (q[:, 2] - z) <= 0) & (np.abs(p[:, 2] - q[:, 2]) > 1e-12)
            t = ((z - p[m, 2]) / (q[m, 2] - p[m, 2]))[:, None]
            pts.append(p[m] + t * (q[m] - p[m]))
        pts = np.concatenate(pts)
        out.append((pts[:, :2].min(0), pts[:, :2].max(0)))
    return out


def hull_slabs(v, kmax=4, tol=0.03, ngrid=48):
    """The collision compound of a convex mesh: its hull cut into <= kmax slabs along z, every slab replaced by the box whose
    footprint is the bounding rectangle of the hull's section at the slab's mid height.  Break heights: the subset of (mesh
    vertex levels + a uniform grid) that minimises the area between the staircase and the true section rectangles; the smallest
    number of slabs whose error is below `tol` of the volume.  Returns (relative error, [(centre, half)])."""
    import itertools
    z0, z1 = v[:, 2].min(), v[:, 2].max()
    fine = z0 + (np.arange(256) + 0.5) / 256 * (z1 - z0)
    sec = _hull_sections(v, fine)
    lo, hi = np.array([s[0] for s in sec]), np.array([s[1] for s in sec])
    area = (hi - lo).prod(1)
    lev = np.unique(np.round(v[:, 2], 4))
    cand = sorted(set(np.round(np.concatenate([lev[(lev > z0 + 1e-4) & (lev < z1 - 1e-4)],
                                               z0 + np.arange(1, ngrid) / ngrid * (z1 - z0)]), 5)))

    def cost(br):
        e, boxes = 0.0, []
        edges = [z0] + list(br) + [z1]
        for a, b in zip(edges[:-1], edges[1:]):
            m = (fine >= a) & (fine < b)
            if not m.any():
                return 1e9, None
            i = int(np.argmin(np.abs(fine - 0.5 * (a + b))))
            bl, bh = lo[i], hi[i]
            inter = np.clip(np.minimum(hi[m], bh) - np.maximum(lo[m], bl), 0, None).prod(1)
            e += (area[m] + (bh - bl).prod() - 2 * inter).sum()
            boxes.append((np.array([(bl[0] + bh[0]) / 2, (bl[1] + bh[1]) / 2, 0.5 * (a + b)]),
                          np.array([(bh[0] - bl[0]) / 2, (bh[1] - bl[1]) / 2, (b - a) / 2])))
        return e / area.sum(), boxes

    best = None
    for k in range(1, kmax + 1):
        best = min((cost(br) for br in itertools.combinations(cand, k - 1)), key=lambda x: x[0])
        if best[0] <= tol:
            break
    return best


# ----------------------------------------------------------------------------- robot
# meshes resolved as slab compounds (file -> number of slabs): the four fingertips (tapered, 26 mm at the base to 19 x 12 mm below
# the rounded end: the bounding box is 1.9 x the hull's volume) and the palm.  31 + 8 + 1 = 40 boxes = SDX_MAX_RBOX
SLAB_MESHES = {"modified_tip.STL": 3, "base_link.STL": 2}


def compile_robot():
    root = ET.parse(ROBOT_URDF).getroot()
    urdf_dir = os.path.dirname(ROBOT_URDF)
    links = {l.get("name"): l for l in root.findall("link")}
    joints = root.findall("joint")
    children = {}
    for j in joints:
        children.setdefault(j.find("parent").get("link"), []).append(j)

    bodies = []  # moving bodies in depth-first URDF order (Isaac Gym's order)

    def shapes_and_inertia(link, R, t):
        """collision boxes + inertial contributions of `link`, expressed through (R, t)
        into the frame of the body it is being folded into."""
        boxes, inert = [], []
        mesh_vol_mass = []
        for c in link.findall("collision"):
            Rc, tc = origin_tf(c.find("origin"))
            Rb, tb = R @ Rc, R @ tc + t
            g = list(c.find("geometry"))[0]
            if g.tag == "box":
                size = np.array([float(v) for v in g.get("size").split()])
                boxes.append({"center": tb.tolist(), "quat": mat_to_quat_xyzw(Rb), "half": (size / 2).tolist(),
                              "src": "urdf_box"})
            elif g.tag == "mesh":
                scale = np.array([float(v) for v in g.get("scale", "1 1 1").split()])
                verts = load_mesh(resolve_mesh(g.get("filename"), urdf_dir)) * scale
                vb = verts @ Rb.T + tb  # in body frame
                lo, hi = vb.min(0), vb.max(0)
                base = os.path.basename(g.get("filename"))
                if base in SLAB_MESHES:      # the shapes that touch bricks: slabs of the hull along the body frame's z axis
                    _, slabs = hull_slabs(vb, kmax=SLAB_MESHES[base], tol=0.0)
                    for sc_, sh_ in slabs:
                        boxes.append({"center": sc_.tolist(), "quat": [0, 0, 0, 1], "half": sh_.tolist(), "src": "mesh_slab:" + base})
                else:
                    boxes.append({"center": ((lo + hi) / 2).tolist(), "quat": [0, 0, 0, 1],
                                  "half": ((hi - lo) / 2).tolist(), "src": "mesh_aabb:" + base})
                vol = ConvexHull(verts).volume
                mesh_vol_mass.append((1000.0 * vol, (lo + hi) / 2, hi - lo))
        ins = link.findall("inertial")
        if ins:
            for i in ins:  # several <inertial> blocks (link_3.0/7.0/11.0): we SUM them (documented)
                m = float(i.find("mass").get("value"))
                Ri, ti = origin_tf(i.find("origin"))
                it = i.find("inertia")
                I = np.array([[float(it.get("ixx")), float(it.get("ixy")), float(it.get("ixz"))],
                              [float(it.get("ixy")), float(it.get("iyy")), float(it.get("iyz"))],
                              [float(it.get("ixz")), float(it.get("iyz")), float(it.get("izz"))]])
                Rw = R @ Ri
                inert.append((m, R @ ti + t, Rw @ I @ Rw.T))
        else:
            for m, c, full in mesh_vol_mass:  # simulator-derived convention: density 1000 x hull volume
                inert.append((m, c, box_inertia(m, full)))
        return boxes, inert

    def fold(link_name, R, t, boxes, inert, names):
        """fold `link_name` and all its fixed-joint descendants into the current body."""
        b, i = shapes_and_inertia(links[link_name], R, t)
        boxes += b
        inert += i
        names.append(link_name)
        moving = []
        for j in children.get(link_name, []):
            Rj, tj = origin_tf(j.find("origin"))
            Rc, tc = R @ Rj, R @ tj + t
            if j.get("type") == "fixed":
                moving += fold(j.find("child").get("link"), Rc, tc, boxes, inert, names)
            else:
                moving.append((j, Rc, tc))
        return moving

    def add_body(link_name, parent_idx, joint, Rj, tj):
        boxes, inert, names = [], [], []
        moving = fold(link_name, np.eye(3), np.zeros(3), boxes, inert, names)
        m = sum(x[0] for x in inert)
        com = sum(x[0] * x[1] for x in inert) / m if m > 0 else np.zeros(3)
        I = np.zeros((3, 3))
        for mi, ci, Ii in inert:
            d = ci - com
            I += Ii + mi * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        idx = len(bodies)
        body = {"name": link_name, "folded": names, "parent": parent_idx, "mass": m, "com": com.tolist(),
                "inertia": I.tolist(), "boxes": boxes}
        if joint is not None:
            ax = np.array([float(v) for v in joint.find("axis").get("xyz").split()])
            lim = joint.find("limit")
            body.update({"joint": joint.get("name"), "joint_pos": tj.tolist(), "joint_quat": mat_to_quat_xyzw(Rj),
                         "axis": (ax / np.linalg.norm(ax)).tolist(),
                         "lower": float(lim.get("lower")), "upper": float(lim.get("upper"))})
        bodies.append(body)
        for j, Rc, tc in moving:
            add_body(j.find("child").get("link"), idx, j, Rc, tc)

    add_body("panda_link0", -1, None, np.eye(3), np.zeros(3))
    assert len(bodies) == 24, len(bodies)
    # drive properties, GS:580-590
    dof = []
    for i in range(23):
        if i < 7:
            dof.append({"kp": 400.0, "kd": 80.0, "effort": 200.0, "vel_limit": [2.175] * 4 + [2.61] * 3})
        else:
            dof.append({"kp": 50.0, "kd": 1.0, "effort": 5.0, "vel_limit": 10.0})
    for i in range(7):
        dof[i]["vel_limit"] = dof[i]["vel_limit"][i]
    return bodies, dof


# ---------------------------------------------------------------------------- bricks
BRICK_NAMES = ['1x2', '1x2_curve', '1x3_curve_soft', '1x3_curve', '1x1', '1x3', '1x4', '2x2_curve_soft']  # GS:706


BODY_TOP = 0.01875        # top of a brick's body = stud base (mesh level, all types); the body bottom is at -BODY_TOP
STUD_SHAFT_TOP = 0.0348   # the stud cylinders (radius 0.0125) end here; a chamfer narrows them to radius 0.009 at the top (0.0387)
STUD_R, STUD_R_TOP = 0.0125, 0.009
CAVITY_ROOF = 0.01        # the underside is hollow up to this level; walls 1.25 - 1.3 mm (mesh levels, all types)


def _stud_centres(v):
    """stud axes of a mesh: the cells of the 30 mm grid that carry vertices of the studs' top level"""
    top = v[np.abs(v[:, 2] - v[:, 2].max()) < 1e-4][:, :2]
    lo, hi = v[:, :2].min(0), v[:, :2].max(0)
    out = []
    for i in range(int(round((hi[0] - lo[0]) / 0.03))):
        for j in range(int(round((hi[1] - lo[1]) / 0.03))):
            c = np.array([lo[0] + 0.03 * (i + 0.5), lo[1] + 0.03 * (j + 0.5)])
            if (np.linalg.norm(top - c, axis=1) < STUD_R_TOP + 1e-3).any():
                out.append(c)
    return out


def compile_bricks():
    out = []
    for name in BRICK_NAMES:
        v = load_stl(os.path.join(REF, "blender/origin_obj", name, name + ".stl")) * 0.01
        lo, hi = v.min(0), v.max(0)
        vol, com, I = hull_mass_props(v)
        mass = 567.0 * vol
        full = hi - lo
        err, slabs = hull_slabs(v, kmax=2, tol=0.03)   # two slabs: every further one multiplies the contacts of a brick pair (<= 4 per pair of boxes)
        # hollow compound (mesh frame): four walls up to the cavity roof, then the hull slabs above the roof
        wall = float(np.abs(v[np.abs(v[:, 2] - CAVITY_ROOF) < 1e-4][:, 1]).max())      # inner half width of the cavity
        t = float(hi[1]) - wall
        zc, zh = (lo[2] + CAVITY_ROOF) / 2, (CAVITY_ROOF - lo[2]) / 2
        hollow = [(np.array([hi[0] - t / 2, 0.0, zc]), np.array([t / 2, full[1] / 2, zh])),
                  (np.array([lo[0] + t / 2, 0.0, zc]), np.array([t / 2, full[1] / 2, zh])),
                  (np.array([0.0, hi[1] - t / 2, zc]), np.array([full[0] / 2 - t, t / 2, zh])),
                  (np.array([0.0, lo[1] + t / 2, zc]), np.array([full[0] / 2 - t, t / 2, zh]))]
        for c, h in slabs:
            a, b = max(c[2] - h[2], CAVITY_ROOF), c[2] + h[2]
            if b > a + 1e-6:
                hollow.append((np.array([c[0], c[1], (a + b) / 2]), np.array([h[0], h[1], (b - a) / 2])))
        out.append({"name": name, "half": (full / 2).tolist(), "center": ((lo + hi) / 2).tolist(), "mass": mass,
                    "com": com.tolist(), "inertia_diag": (567.0 * np.diag(I)).tolist(), "inertia_offdiag_xz": float(567.0 * I[0, 2]),
                    "inertia_diag_bbox": np.diag(box_inertia(mass, full)).tolist(), "hull_volume": vol, "slab_error": float(err),
                    "sub": [{"center": c.tolist(), "half": h.tolist()} for c, h in slabs],
                    "hollow": [{"center": c.tolist(), "half": h.tolist()} for c, h in hollow], "wall": t})
    v = load_stl(os.path.join(REF, "blender/assets_for_insertion/origin_obj/4x4x1_real/4x4x1_real.stl")) * 0.01
    lo, hi = v.min(0), v.max(0)
    plate = {"name": "4x4x1_real", "half": ((hi - lo) / 2).tolist(), "center": ((lo + hi) / 2).tolist(), "sub": plate_compound(v)}
    return out, plate


def plate_compound(v):
    """a base plate as body + studs (mesh frame): the body box up to the stud base, every stud as its shaft (the square around the
    cylinder) and its chamfered tip (the square around the mean of the chamfer's radii)"""
    lo, hi = v.min(0), v.max(0)
    top = float(hi[2]) - (0.0387 - BODY_TOP)                              # stud base of THIS plate (the 4x4x2 / x4 plates are taller)
    sub = [{"center": [float(lo[0] + hi[0]) / 2, float(lo[1] + hi[1]) / 2, (top + float(lo[2])) / 2],
            "half": [float(hi[0] - lo[0]) / 2, float(hi[1] - lo[1]) / 2, (top - float(lo[2])) / 2]}]
    s1, s2 = STUD_SHAFT_TOP - BODY_TOP, 0.0387 - STUD_SHAFT_TOP
    rt = 0.5 * (STUD_R + STUD_R_TOP)
    for c in _stud_centres(v):
        sub.append({"center": [float(c[0]), float(c[1]), top + s1 / 2], "half": [STUD_R, STUD_R, s1 / 2]})
        sub.append({"center": [float(c[0]), float(c[1]), top + s1 + s2 / 2], "half": [rt, rt, s2 / 2]})
    return sub


def compile_insert_plates():
    """InsertSim's three base plates (IS:750-767, env % 3) as stud compounds.  A brick seats with its origin 0.0375 (1 + k) above the
    plate origin (IS:1123-1125): its walls stand on the plate BODY, the plate's studs inside its hollow underside."""
    out = []
    for name in ["4x4x1_real", "4x4x2_real", "4x4x4_real"]:
        v = load_stl(os.path.join(REF, "blender/assets_for_insertion/origin_obj", name, name + ".stl")) * 0.01
        lo, hi = v.min(0), v.max(0)
        sub = plate_compound(v)
        out.append({"name": name, "half": ((hi - lo) / 2).tolist(), "center": ((lo + hi) / 2).tolist(),
                    "bbox_lo": lo.tolist(), "bbox_hi": hi.tolist(), "sub": sub})
    return out


def main():
    bodies, dof = compile_robot()
    bricks, plate = compile_bricks()

    # ---- static boxes (world frame), GS:629-685, 748-808, 827-838
    statics = []

    def sbox(name, c, full, friction=1.0):
        statics.append({"name": name, "center": list(c), "half": [x / 2 for x in full], "quat": [0, 0, 0, 1]})

    sbox("table", (0.0, 0.0, 0.3), (1.5, 1.0, 0.6))
    bx, by, bz, th, ox, oy = 0.60, 0.416, 0.165, 0.01, 0.25, 0.19
    sbox("bin_bottom", (ox, oy, 0.6 + th / 2), (bx, by, th))
    sbox("bin_left", (ox, (by - th) / 2 + oy, 0.6 + bz / 2), (bx, th, bz))
    sbox("bin_right", (ox, -(by - th) / 2 + oy, 0.6 + bz / 2), (bx, th, bz))
    sbox("bin_former", ((bx - th) / 2 + ox, oy, 0.6 + bz / 2), (th, by, bz))
    sbox("bin_after", (-(bx - th) / 2 + ox, oy, 0.6 + bz / 2), (th, by, bz))
    # 10 rows x 6 fixed bricks (types 0/5/6, half lengths .03/.045/.06, 6 mm gaps) at z=0.625:
    # row j spans x in [0.504 - 0.54, 0.504], y = 0.365 - 0.039 j +- 0.015  -> one slab.
    b0 = bricks[0]
    zc, zh = 0.625 + b0["center"][2], b0["half"][2]
    x_hi = 0.254 + 0.25
    x_lo = x_hi - (2 * (3 * 0.03 + 0.045 + 2 * 0.06) + 5 * 0.006)
    y_hi = 0.175 + 0.19 + b0["half"][1]
    y_lo = 0.175 + 0.19 - 0.039 * 9 - b0["half"][1]
    sbox("brick_floor", ((x_lo + x_hi) / 2, (y_lo + y_hi) / 2, zc), (x_hi - x_lo, y_hi - y_lo, 2 * zh))
    pc = np.array([0.25, -0.19, 0.618]) + np.array(plate["center"])
    sbox("base_plate", pc.tolist(), [2 * h for h in plate["half"]])      # bounding box; the stud compound is scene["base_plate"]["sub"]

    # fixed-brick root states (actor order inside the 132-brick block: 72 free then 60 fixed), deterministic
    # stand-in for the build-time random.shuffle at GS:755-759: pattern [0,0,0,1,2,2] rotated by row index.
    fixed_bricks = []
    lego_list, bian = [0, 5, 6], [0.03, 0.045, 0.06]
    for j in range(10):
        ran = [0, 0, 0, 1, 2, 2]
        ran = ran[j % 6:] + ran[:j % 6]
        cx = 0.254 - bian[ran[0]] + 0.25
        cy = 0.175 + 0.19 - 0.039 * j
        for k in range(6):
            fixed_bricks.append({"type": lego_list[ran[k]], "pos": [cx, cy, 0.625]})
            if k < 5:
                cx -= bian[ran[k]] + bian[ran[k + 1]] + 0.006

    # free brick spawn lattice GS:737-742 (yaw 0.785 about z)
    free_spawn = []
    for n in range(9):
        for i in range(8):
            if n % 2 == 0:
                p = [-0.17 + 0.17 * (i % 3) + 0.25, -0.11 + 0.11 * (i // 3) + 0.19, 0.62 + n * 0.06]
            else:
                p = [0.17 - 0.17 * (i % 3) + 0.25, 0.11 - 0.11 * (i // 3) + 0.19, 0.62 + n * 0.06]
            free_spawn.append({"type": i, "pos": p, "quat": [0.0, 0.0, math.sin(0.785 / 2), math.cos(0.785 / 2)]})

    # camera offset, GS:887-889: Quat.from_euler_zyx(0, -3.141+0.5, 1.571) -> args are (x=roll, y=pitch, z=yaw)
    cam_q = mat_to_quat_xyzw(rpy_to_mat(0.0, -3.141 + 0.5, 1.571))

    scene = {
        "format": "seqdex_amd.scene.v2",
        "provenance": "derived by tools/compile_scene.py from reference assets (URDF/STL/OBJ); numbers only",
        "robot": {"base_pos": [-0.35, 0.0, 0.6], "base_quat": [0, 0, 0, 1], "bodies": bodies, "dof": dof,
                  "hand_base_body": 7, "fingertip_bodies": None, "arm_contact_bodies": [1, 2, 3, 4, 5, 6]},
        "brick_types": bricks, "base_plate": plate, "base_plate_pos": [0.25, -0.19, 0.618],
        "insert_plates": compile_insert_plates(), "insert_plate_pos": [0.25, -0.2, 0.618],      # IS:1438-1440
        "statics": statics, "fixed_bricks": fixed_bricks, "free_spawn": free_spawn,
        "camera_offset_quat": cam_q, "camera_offset_pos": [0.03, 0.107 - 0.098, 0.067 + 0.107],
        "vestigial_object_pos": [0.0, 0.0, -10.78], "vestigial_goal_pos": [-0.2, -0.06, -10.78 - 10.12 - 0.04],
        "sim": {"dt": 1.0 / 60.0, "substeps": 2, "pos_iters": 16, "contact_offset": 0.002, "gravity": [0, 0, -9.81]},
    }
    names = [b["name"] for b in bodies]
    scene["robot"]["fingertip_bodies"] = [names.index(n) for n in ["link_3.0", "link_7.0", "link_11.0", "link_15.0"]]
    assert names.index("panda_link7") == 7
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(scene, f, indent=1)
    print("wrote", os.path.normpath(OUT))
    for i, b in enumerate(bodies):
        print(i, b["name"], "parent", b["parent"], "m=%.4f" % b["mass"], "nbox", len(b["boxes"]), b.get("joint"))
    for b in bricks:
        print(b["name"], "half", np.round(b["half"], 4), "c", np.round(b["center"], 4), "m=%.4f" % b["mass"], "com", np.round(b["com"], 4),
              "slabs", len(b["sub"]), "err %.3f" % b["slab_error"], "hollow", len(b["hollow"]), "Ixz/Ixx %.2f" % (b["inertia_offdiag_xz"] / b["inertia_diag"][0]))
    for s in statics:
        print(s["name"], np.round(s["center"], 4), np.round(s["half"], 4))


if __name__ == "__main__":
    sys.exit(main())
