#!/usr/bin/env python3
"""find_anchor_sizes.py -- cluster the (height, width) of every annotated box to suggest YOLO anchors.
Reference: find_anchor_sizes.py:19-66 (same flag, same printed lines: a score and the centres for k = 2..7).

The reference delegates to sklearn.cluster.KMeans (unseeded) and saves matplotlib scatter plots.  This is an offline
host utility outside the GPU path, so it is plain NumPy: k-means++ seeding + Lloyd iterations, best of `n_init`
restarts, `score` = minus the within-cluster sum of squares (what KMeans.score returns).  Being unseeded in the
reference, its centres are only defined up to the usual k-means local optima; here the seed is a flag.  Plots are
written only when matplotlib is importable."""
import argparse
import os

import numpy as np

from yolo3 import bbox_utils


def kmeans(X, k, rng, n_init=10, max_iter=300, tol=1e-4):
    """-> (centers [k,2], labels [n], inertia).  k-means++ init, Lloyd, best of n_init."""
    X = np.asarray(X, np.float64)
    n = X.shape[0]
    if n < k:
        raise ValueError('need at least %d boxes for %d clusters, got %d' % (k, k, n))
    best = None
    scale = tol * float(np.mean(np.var(X, axis=0))) if n > 1 else 0.0       # sklearn's tolerance scaling
    for _ in range(n_init):
        centers = np.empty((k, X.shape[1]))
        centers[0] = X[rng.integers(n)]
        d2 = ((X - centers[0]) ** 2).sum(1)
        for j in range(1, k):
            tot = d2.sum()
            idx = rng.integers(n) if tot <= 0 else int(np.searchsorted(np.cumsum(d2), rng.random() * tot))
            centers[j] = X[min(idx, n - 1)]
            d2 = np.minimum(d2, ((X - centers[j]) ** 2).sum(1))
        for _ in range(max_iter):
            dist = ((X[:, None, :] - centers[None, :, :]) ** 2).sum(2)
            labels = dist.argmin(1)
            new = centers.copy()
            for j in range(k):
                m = labels == j
                if m.any():
                    new[j] = X[m].mean(0)
            shift = float(((new - centers) ** 2).sum())
            centers = new
            if shift <= scale:
                break
        dist = ((X[:, None, :] - centers[None, :, :]) ** 2).sum(2)
        labels = dist.argmin(1)
        inertia = float(dist[np.arange(n), labels].sum())
        if best is None or inertia < best[2]:
            best = (centers, labels, inertia)
    return best


def load_sizes(csv_dirpath):
    """[n,2] = (H, W) of every box of every csv in the folder (find_anchor_sizes.py:20-30)."""
    h_list, w_list = [], []
    for fn in sorted(f for f in os.listdir(csv_dirpath) if f.endswith('.csv')):
        boxes = bbox_utils.load_boxes_to_xywhc(os.path.join(csv_dirpath, fn))
        w_list.extend(boxes[:, 2].tolist())
        h_list.extend(boxes[:, 3].tolist())
    return np.stack([np.asarray(h_list, np.float64), np.asarray(w_list, np.float64)], axis=1).reshape(-1, 2)


def find_anchors(csv_dirpath, seed=0, plot=True):
    X = load_sizes(csv_dirpath)
    rng = np.random.default_rng(seed)
    plt = None
    if plot:
        try:
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
        except ImportError:
            print('matplotlib is not installed: scatter plots skipped')
    out = {}
    for k in range(2, 8):
        centers, labels, inertia = kmeans(X, k, rng)
        out[k] = centers
        print('score for {}-means = {}'.format(k, -inertia))
        print('  centers = {}'.format(centers))
        if plt is not None:
            fig = plt.figure(figsize=(16, 9), dpi=200)
            plt.scatter(X[:, 0], X[:, 1], c=labels, cmap='viridis')
            plt.xlabel('Width')           # axis labels as in the reference (its columns are H, W)
            plt.ylabel('Height')
            plt.scatter(centers[:, 0], centers[:, 1], c='black', s=200, alpha=0.5)
            plt.savefig('scatterplot_{}_clusters.png'.format(k))
            plt.close(fig)
        print('View the scatterplot and determine if the clusters look appropriate. You generally want a small, medium, and large anchor for Yolo.')
    return out


if __name__ == '__main__':
    parser = argparse.ArgumentParser(prog='find_anchor_sizes', description='Script to determine what anchors to use with yolov3.')
    parser.add_argument('--csv_dirpath', dest='csv_dirpath', type=str,
                        help='Filepath to the directory containing annotation csv files with columns [X,Y,W,H]', required=True)
    parser.add_argument('--seed', type=int, default=0, help='k-means seed (extension; the reference is unseeded)')
    args = parser.parse_args()
    find_anchors(args.csv_dirpath, args.seed)
