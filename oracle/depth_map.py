"""TEST INFRASTRUCTURE ONLY -- CPU oracle, never imported by the product path.

Restatement of the reference's common/drop_depth_map.py (class DropDepthMap): scene depth -> camera-frame XYZ through
KITTI's rectified camera-2 projection, and the distance of every scene point from a set of drop start positions.  In the
reference this is dead code (constructed only behind USE_DEPTH_WEIGHTING = 0, generator.py:20,339-341), so no output of
the hot path depends on it; it is restated here -- as three plain functions, pinned bit for bit against the reference's
own class by tests/golden/reference_vectors.npz (make_golden.py section 8d) -- because BASELINE.json's north_star
names it.  The depth test the product actually offers is the default-off RR_OPT_DEPTH_OCCLUSION option, defined in
oracle/render.py (_visible).

The matrix products below go through np.dot with the reference's operand shapes on purpose: BLAS summation order is
part of the bits being pinned.
"""
import numpy as np

CAMERA_HEIGHT_M = 1.65          # drop_depth_map.py:35: camera 0 sits 1.65 m above the ground plane


def read_calibration(path):
    """drop_depth_map.py:21-52 -> dict(P_R = P_rect_02 @ [R_rect_02 | 0; 0 | 1], P_R_pinv, camera_pos_world).
    Lines are `key: v v v ...` (KITTI raw calib_cam_to_cam.txt); only P_rect_02 and R_rect_02 are read."""
    P = R = None
    for line in open(path, 'r').read().split('\n'):
        key = line[0:10]
        if key in ('P_rect_02:', 'R_rect_02:'):
            vals = np.array(line.split(':')[1].split(' ')[1:]).astype(float)
            if key[0] == 'P':
                P = vals.reshape((3, 4))
            else:
                R = vals.reshape((3, 3))
    R44 = np.identity(4).astype(float)
    R44[:3, :3] = R
    ground = np.array([0., CAMERA_HEIGHT_M, 0.0]).reshape((3, 1))
    cam2_wrt_cam0 = np.zeros((3, 1))
    cam2_wrt_cam0[0] = P[0, 3] / (-P[0, 0])                    # camera 2 is shifted along x (:42-43)
    P_R = np.dot(P, R44)
    return dict(P_R=P_R, P_R_pinv=np.linalg.pinv(P_R), camera_pos_world=cam2_wrt_cam0 + (-ground))


def backproject(depth_map, P_R_pinv):
    """drop_depth_map.py:54-86 (return_xyz + get_world_points): H x W x 3 camera-frame points, y negated (:84).
    (The reference reshapes to a hard-coded 352 x 1216, :70; here: the depth map's own size.)"""
    H, W = depth_map.shape[:2]
    xx, yy = np.meshgrid(np.arange(W), np.arange(H))
    pix = np.concatenate((xx[..., None], yy[..., None], np.ones((H, W, 1))), axis=-1)
    xyz = np.dot(P_R_pinv, pix.reshape((-1, 3)).T).T.reshape((H, W, 4))[:, :, :3]
    xyz *= (depth_map / xyz[:, :, 2])[..., None]
    xyz[:, :, 1] = -xyz[:, :, 1]
    return xyz


def drop_distance_maps(drops_start, xyz_map):
    """drop_depth_map.py:88-97 (depth_map_drop): (N, H, W) float16 distances |scene point - drop start|."""
    out = np.zeros((drops_start.shape[0],) + xyz_map.shape[:2]).astype(np.float16)
    d = np.reshape(drops_start, (-1, 1, 1, 3))
    m = xyz_map[None]
    out[:, :, :] = np.sqrt(np.square(m[..., 0] - d[..., 0]) + np.square(m[..., 1] - d[..., 1]) + np.square(m[..., 2] - d[..., 2]))
    return out
