"""TEST INFRASTRUCTURE ONLY (CPU oracle).  numpy restatement of the reference scan loaders:
read_pc of datasets/mulran/mulran_raw.py:19-25 / datasets/kitti/kitti_raw.py:16-22 and the preprocessing of
PointCloudLoader.__call__, misc/point_clouds.py:95-111 (that module itself imports open3d and cannot be imported)."""
import numpy as np

GROUND_PLANE_LEVEL = {"mulran": -0.9, "kitti": -1.5, "southbay": -1.6}


def read_pc(raw_bytes_or_array):
    pc = np.frombuffer(raw_bytes_or_array, dtype=np.float32) if isinstance(raw_bytes_or_array, (bytes, bytearray)) \
        else np.asarray(raw_bytes_or_array, dtype=np.float32).reshape(-1)
    return np.reshape(pc, (-1, 4))[:, :3]                                    # mulran_raw.py:22-24


def preprocess(pc, dataset_type="mulran", remove_zero_points=True, remove_ground_plane=True):
    if remove_zero_points:
        mask = np.all(np.isclose(pc, 0), axis=1)                             # point_clouds.py:103-105
        pc = pc[~mask]
    if remove_ground_plane:
        mask = pc[:, 2] > GROUND_PLANE_LEVEL[dataset_type]                   # :107-109
        pc = pc[mask]
    return pc
