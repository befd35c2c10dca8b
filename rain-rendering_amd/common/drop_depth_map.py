"""Mirror of the reference's common/drop_depth_map.py (DropDepthMap: pixel depth -> camera-frame XYZ through KITTI's
rectified projection, and per-drop distance maps).  In the reference this class is only ever constructed behind
``USE_DEPTH_WEIGHTING = 0`` (generator.py:20,339-341): dead code, no output depends on it.  It is kept for API parity
(same methods, same arithmetic; pinned against the reference's own functions by tests/golden) and as the host-side
description of what the library's default-off ``RR_OPT_DEPTH_OCCLUSION`` option tests on the device: a drop farther
from the camera than the scene at a pixel is not composited there.

Differences: `return_xyz` reshapes to the depth map's own size (the reference hard-codes 352 x 1216, :70)."""
import numpy as np


class DropDepthMap:
    def __init__(self, filename):                     # reference drop_depth_map.py:13-19
        self.filename = filename
        self.P2_R_rect = None
        self.P2_R_inv = None
        self.camera_pos_world = None
        self.world_pts_acc_cam = None
        self.camera_pos_wrt_cam0 = None

    def get_util_matrices(self):                      # reference drop_depth_map.py:21-52
        with open(self.filename, 'r') as fh:
            lines = fh.read().split('\n')
        P2_rect = R2_rect = None
        for line in lines:
            if line[0:10] == 'P_rect_02:':
                P2_rect = np.array(line.split(':')[1].split(' ')[1:]).astype(float).reshape((3, 4))
            elif line[0:10] == 'R_rect_02:':
                R2_rect = np.array(line.split(':')[1].split(' ')[1:]).astype(float).reshape((3, 3))
        R2_rect_44 = np.identity(4).astype(float)
        R2_rect_44[:3, :3] = R2_rect
        self.world_pts_acc_cam = np.array([0., 1.65, 0.0]).reshape((3, 1))     # camera 1.65 m above the ground
        camera_pos_wrt_world = - self.world_pts_acc_cam
        self.camera_pos_wrt_cam0 = np.zeros((3, 1))
        self.camera_pos_wrt_cam0[0] = P2_rect[0, 3] / (-P2_rect[0, 0])
        self.camera_pos_world = self.camera_pos_wrt_cam0 + camera_pos_wrt_world
        self.P2_R_rect = np.dot(P2_rect, R2_rect_44)
        self.P2_R_inv = np.linalg.pinv(self.P2_R_rect)

    def return_xyz(self, depth_map):                  # reference drop_depth_map.py:54-76
        x = np.arange(depth_map.shape[1])
        y = np.arange(depth_map.shape[0])
        z = np.ones((depth_map.shape[0], depth_map.shape[1], 1))
        xx, yy = np.meshgrid(x, y)
        xx = np.expand_dims(xx, axis=-1)
        yy = np.expand_dims(yy, axis=-1)
        xyz_image = np.concatenate((xx, yy, z), axis=-1)
        xyz_coord = np.dot(self.P2_R_inv, xyz_image.reshape((-1, 3)).T).T
        xyz_coord = np.reshape(xyz_coord, (depth_map.shape[0], depth_map.shape[1], 4))
        xyz_coord = xyz_coord[:, :, :3]
        scale_term = depth_map / xyz_coord[:, :, 2]
        scale_term = np.expand_dims(scale_term, axis=-1)
        xyz_coord *= scale_term
        return xyz_coord

    def get_world_points(self, depth_map):            # reference drop_depth_map.py:78-86
        self.get_util_matrices()
        xyz_coord = self.return_xyz(depth_map)
        xyz_coord[:, :, 1] = -xyz_coord[:, :, 1]
        return xyz_coord

    @staticmethod
    def depth_map_drop(drops_start, xyz_map):         # reference drop_depth_map.py:88-97
        """(N, H, W) float16: distance between the scene point behind every pixel and every drop's start."""
        depth_maps = np.zeros((drops_start.shape[0], xyz_map.shape[0], xyz_map.shape[1])).astype(np.float16)
        ds = np.reshape(drops_start, (-1, 1, 1, 3))
        depth_maps[:, :, :] = np.sqrt(np.square(xyz_map[None, :, :, 0] - ds[:, :, :, 0]) +
                                      np.square(xyz_map[None, :, :, 1] - ds[:, :, :, 1]) +
                                      np.square(xyz_map[None, :, :, 2] - ds[:, :, :, 2]))
        return depth_maps
