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
  * per-link collision boxes: URDF <box> shapes verbatim, meshes replaced by
    their link-frame axis-aligned bounding box (DESIGN.md §3 "all shapes are boxes");
  * PD drive gains / effort / velocity limits [GS:580-590];
  * the 8 brick types: bounding box, box centre offset, mass = 567 x hull volume
    [assets/urdf/blender/urdf/*.urdf], and the base plate box;
  * static scene boxes: table, 5 bin walls [GS:629-685], the 60-brick fixed floor
    merged into one slab [GS:748-808], base plate [GS:827-838];
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


# ----------------------------------------------------------------------------- robot
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
                boxes.append({"center": ((lo + hi) / 2).tolist(), "quat": [0, 0, 0, 1],
                              "half": ((hi - lo) / 2).tolist(), "src": "mesh_aabb:" + os.path.basename(g.get("filename"))})
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


def compile_bricks():
    out = []
    for name in BRICK_NAMES:
        v = load_stl(os.path.join(REF, "blender/origin_obj", name, name + ".stl")) * 0.01
        lo, hi = v.min(0), v.max(0)
        vol = ConvexHull(v).volume
        mass = 567.0 * vol
        full = hi - lo
        out.append({"name": name, "half": (full / 2).tolist(), "center": ((lo + hi) / 2).tolist(), "mass": mass,
                    "inertia_diag": np.diag(box_inertia(mass, full)).tolist(), "hull_volume": vol})
    v = load_stl(os.path.join(REF, "blender/assets_for_insertion/origin_obj/4x4x1_real/4x4x1_real.stl")) * 0.01
    lo, hi = v.min(0), v.max(0)
    plate = {"name": "4x4x1_real", "half": ((hi - lo) / 2).tolist(), "center": ((lo + hi) / 2).tolist()}
    return out, plate


def compile_insert_plates():
    """InsertSim's three base plates (IS:750-767, env % 3).  A brick seats with its origin 0.0375 (1 + k) above the plate origin
    (IS:1123-1125), i.e. its body rests on the plate BODY; the studs that the V-HACD hulls engage are not representable by a box, so
    the collision box is the body without the stud layer (stud height = the bricks' own: bounding-box top minus the 0.01875 body top)."""
    stud = None
    out = []
    for name in ["4x4x1_real", "4x4x2_real", "4x4x4_real"]:
        v = load_stl(os.path.join(REF, "blender/assets_for_insertion/origin_obj", name, name + ".stl")) * 0.01
        lo, hi = v.min(0), v.max(0)
        if stud is None:
            stud = float(hi[2] - 0.01875)
        top = float(hi[2]) - stud
        out.append({"name": name, "half": [float(hi[0] - lo[0]) / 2, float(hi[1] - lo[1]) / 2, (top - float(lo[2])) / 2],
                    "center": [float(lo[0] + hi[0]) / 2, float(lo[1] + hi[1]) / 2, (top + float(lo[2])) / 2],
                    "bbox_lo": lo.tolist(), "bbox_hi": hi.tolist()})
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
    sbox("base_plate", pc.tolist(), [2 * h for h in plate["half"]])

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
        "format": "seqdex_amd.scene.v1",
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
        print(b["name"], "half", np.round(b["half"], 4), "c", np.round(b["center"], 4), "m=%.4f" % b["mass"])
    for s in statics:
        print(s["name"], np.round(s["center"], 4), np.round(s["half"], 4))


if __name__ == "__main__":
    sys.exit(main())
