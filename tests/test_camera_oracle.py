"""CPU: known-answer checks of oracle/camera_oracle.py (the numpy ray caster the segmentation-camera kernel is compared with):
a lone brick on the optical axis projects to the analytic pinhole footprint, nearer boxes occlude farther ones, ids follow the
brick index + 1 rule (SE:840)."""
import numpy as np

from oracle import camera_oracle as CO


def _scene_with_one_brick(scene, pos, quat=(0, 0, 0, 1), brick=0):
    d = scene.to_desc(task_kind=3)
    d.n_static = 0
    d.n_rbox = 0
    root = np.zeros((142, 13), np.float32)
    root[:, 6] = 1
    root[9:141, 0:3] = [50.0, 50.0, -50.0]          # everything else far outside the view
    root[9 + brick, 0:3] = pos
    root[9 + brick, 3:7] = quat
    return d, root, np.zeros((165, 13), np.float32)


def test_lone_brick_footprint_and_centroid(scene):
    # camera at (0.35, 0.19, 1.0) looking at (0.2, 0.19, 0): put brick 0 on the axis, 0.5 m from the camera
    cam, tgt = np.array([0.35, 0.19, 1.0]), np.array([0.2, 0.19, 0.0])
    f = (tgt - cam) / np.linalg.norm(tgt - cam)
    d, root, rb = _scene_with_one_brick(scene, cam + 0.5 * f)
    img = CO.render(d, root, rb)
    assert set(np.unique(img)) == {0, 1}
    n, cx, cy = CO.pixel_stats(img, 1)
    assert abs(cx - 63.5) <= 2 and abs(cy - 63.5) <= 2                       # on the optical axis
    # a 0.06 x 0.03 x 0.057 box at 0.5 m under a 90-degree field of view: 64 pixels per metre at unit depth -> 128 px/m at 0.5 m
    h = np.array(list(d.brick_half[d.brick_type[0]]))
    lo = (2 * h[0] * 128) * (2 * h[1] * 128) * 0.5                          # at least half the top-face footprint
    hi = (2 * np.linalg.norm(h) * 128) ** 2                                  # at most the bounding-sphere footprint
    assert lo < n < hi, (n, lo, hi)


def test_nearer_box_occludes_and_ids_follow_brick_index(scene):
    cam, tgt = np.array([0.35, 0.19, 1.0]), np.array([0.2, 0.19, 0.0])
    f = (tgt - cam) / np.linalg.norm(tgt - cam)
    d, root, rb = _scene_with_one_brick(scene, cam + 0.6 * f, brick=5)
    img_far_only = CO.render(d, root, rb)
    assert CO.pixel_stats(img_far_only, 6)[0] > 0                             # brick 5 carries id 6
    root[9 + 70, 0:3] = cam + 0.3 * f                                        # brick 70 (id 71) in front of it
    img = CO.render(d, root, rb)
    assert CO.pixel_stats(img, 71)[0] > CO.pixel_stats(img_far_only, 6)[0]    # nearer = larger
    assert CO.pixel_stats(img, 6)[0] < CO.pixel_stats(img_far_only, 6)[0]     # and hides part of the farther one
    assert CO.pixel_stats(img, 3) == (0, 0, 0)                                # an id that is not in view
