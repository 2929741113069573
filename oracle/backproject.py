"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's back-projection
(``pointstowood/src/predicter.py:107-142``, ``PointCloudClassifier``).

``compute_labels`` (:112-127) is restated statement by statement in numpy (the reference runs it through numba).
The neighbour search is third-party there (pykdtree ``KDTree.query``, not vendored, not installed here): it is
restated with scipy's cKDTree over float64 coordinates.  PINNED: ``compute_labels`` and ``collect_predictions`` are
checked against outputs of the reference's own ``PointCloudClassifier`` (imported with numba.jit -> identity and
pykdtree -> the same scipy shim; tests/golden/make_golden_host.py -> tests/golden/host/vote.npz, collect.npz,
segmentation.npz) by tests/test_host_golden.py.  What stays unpinned is only pykdtree's tie order among equidistant
neighbours, which no published contract fixes.
"""
import numpy as np


def knn_indices(classified_xyz, original_xyz, k):
    """predicter.py:136-137: kd_tree.query(original[:, :3], k)."""
    from scipy.spatial import cKDTree
    tree = cKDTree(np.asarray(classified_xyz, dtype=np.float64))
    _, idx = tree.query(np.asarray(original_xyz, dtype=np.float64), k=k)
    return idx.reshape(len(original_xyz), k)


def compute_labels(nbr_classification, any_wood):
    """predicter.py:112-127.  nbr_classification [n, k, 5] (x, y, z, pred, prob) float64 -> labels [n, 2]."""
    n, num_classes = nbr_classification.shape[0], nbr_classification.shape[1]   # :115-116 (k, sic)
    labels = np.zeros((n, 2))
    for i in range(n):
        labels[i, 1] = np.median(nbr_classification[i, :, -1])                 # :118
        if any_wood != 1:                                                        # :119-121
            labels[i, 0] = 1 if np.any(nbr_classification[i, :, -2] > any_wood) else 0
        else:                                                                    # :122-126
            votes = np.zeros(num_classes)
            for j in range(num_classes):
                votes[j] = np.sum((nbr_classification[i, :, -2] == j) * nbr_classification[i, :, -1])
            labels[i, 0] = np.argmax(votes)
    return labels


def collect_predictions(classification, original_xyz, any_wood):
    """predicter.py:129-142 without the DataFrame / nbrs.npy cache: classification [nc, 5] float64."""
    idx = knn_indices(classification[:, :3], original_xyz, 32 if any_wood != 1 else 64)
    return compute_labels(classification[idx], any_wood)
