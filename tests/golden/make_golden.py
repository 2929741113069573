#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz FROM THE REFERENCE ITSELF.

Runs only in the authoring container: it puts ``oracle/stubs`` (our own re-exports of
``oracle/ops.py`` under the torch_geometric / torch_scatter module paths, which are
not installed here) and ``/root/reference/pointstowood`` on ``sys.path``, imports the
reference's unmodified ``src.model.Net`` (which pulls in ``src.pointnet``), loads a
recipe-generated checkpoint (``pointstowood_amd/synthetic_weights.py``; the trained ``global.pth`` is
absent from the mount) and records inputs, per-level sample indices, neighbour
lists, level outputs and logits.  The reference's source never enters this repo;
only these data vectors do.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz + manifest.json
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.insert(0, "/root/reference/pointstowood")

import src.model as ref_model  # noqa: E402  (the reference, over the stubs)
from pointstowood_amd import synthetic_voxels as synth, synthetic_weights as weights  # noqa: E402

FULL_LIMIT = 50_000  # tensors with more elements are stored as checksum + sampled rows
SAMPLE_ROWS = 16


class _Data:
    pass


def _cases():
    U, S = synth.uniform_voxel, synth.surface_voxel
    dup = U(2.0, 1500, 11, True)
    dup["pos"][700:1400] = dup["pos"][0:700]          # exact duplicates -> distance ties
    dup["reflectance"][700:1400] = dup["reflectance"][0:700]
    tiny = U(0.3, 140, 21, True)                      # <32 points at levels 2/3 -> knn returns < k
    return [
        # name, voxels, C, k, weight seed
        ("u2_2k_k16_c32", [U(2.0, 2048, 123, False)], 32, 16, 0),          # BASELINE config 1
        ("ragged_b2_refl_c8", [U(2.0, 4096, 124, True), U(2.0, 512, 125, True)], 8, 32, 0),
        ("surface_cap_c4", [S(2.0, 3000, 7, True)], 4, 32, 0),
        ("dups_tiny_b3_c4", [dup, tiny, U(2.0, 128, 31, False)], 4, 32, 1),
        ("u4_3k_refl_c8", [U(4.0, 3000, 126, True)], 8, 32, 1),
        ("u2_16k_c32", [U(2.0, 16384, 123, False)], 32, 32, 0),             # canonical U2-16k
        # BASELINE configs[1] at its own size - the bench workload's batch 0 (bench.py host_batch(0, 0)): B = 8 x U(2.0, 16 384,
        # seeds 123..130), k = 32, xyz only.  Inputs are regenerated from the recipe by the tests (checksums + samples here);
        # every logit is stored (0.5 MB), the level tensors as checksums + sampled rows.  ~1 min of reference forward on 8 vCPUs.
        ("config1_b8_16k_c32", [U(2.0, 16384, 123 + i, False) for i in range(8)], 32, 32, 0,
         {"full": ("logits",), "recipe": [("uniform", 2.0, 16384, 123 + i, 0) for i in range(8)]}),
        # the plot regime (configs[3]): 8 voxels of 2 m cells cut out of the synthetic forest plot - stems, crowns and ground:
        # far more level-2 / 3 points per input point than a uniform cube (M3 / N ~ 0.5)
        ("forest_plot_b8_c32", plot_voxels(), 32, 32, 0, {"full": ("logits",)}),
    ]


def plot_voxels(n=400_000, side=20.0, cell=2.0, count=8):
    """`count` voxels of a `cell`-metre grid over synth.forest_plot(n, side) with 128..16384 points, spread over the size range
    (plain floor binning - the voxeliser's own arithmetic is pinned elsewhere: tests/golden/make_golden_host.py), each centred and
    scaled like TestingDataset.__getitem__ (synth._finish); reflectance squashed into (-1, 1)."""
    pc = synth.forest_plot(n, seed=3, side=side)
    key = torch.floor(pc[:, :3] / cell).to(torch.long)
    key = (key[:, 0] * 4096 + key[:, 1]) * 4096 + key[:, 2]
    order = torch.argsort(key, stable=True)
    _, counts = torch.unique_consecutive(key[order], return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    ok = ((counts >= 128) & (counts <= 16384)).nonzero(as_tuple=True)[0]
    ok = ok[torch.argsort(counts[ok], stable=True)]
    pick = ok[torch.linspace(0, len(ok) - 1, count).round().long()]
    out = []
    for v in pick.tolist():
        rows = order[starts[v]: starts[v] + counts[v]]
        out.append(synth._finish(pc[rows, :3].contiguous(), torch.tanh(pc[rows, 3] / 10.0).contiguous()))
    return out


def _store(out, name, t, full=False):
    a = t.detach().cpu().numpy()
    if a.dtype == np.int64:
        a = a.astype(np.int32)
    if a.size <= FULL_LIMIT or full:
        out[name] = a
    else:
        rows = np.linspace(0, a.shape[0] - 1, SAMPLE_ROWS).astype(np.int64)
        out[name + "__rows"] = rows.astype(np.int32)
        out[name + "__sample"] = a[rows]
        out[name + "__shape"] = np.array(a.shape, dtype=np.int64)
        if a.dtype.kind == "f":
            out[name + "__sum"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
        else:
            out[name + "__sum"] = np.array([a.astype(np.int64).sum(), (a.astype(np.int64) * (np.arange(a.size).reshape(a.shape) % 8191 + 1)).sum()])


def run_case(name, voxels, C, k, wseed, opts=None):
    opts = opts or {}
    torch.manual_seed(0)
    net = ref_model.Net(num_classes=1, C=C).eval()
    tab = weights.key_table(1, C)
    ref_sd = net.state_dict()
    assert [t[0] for t in tab] == list(ref_sd.keys()), "state-dict key order differs from the reference"
    for key, shape, _ in tab:
        assert tuple(ref_sd[key].shape) == tuple(shape), key
    net.load_state_dict(weights.synth_state_dict(1, C, seed=wseed), strict=True)
    for m in (net.sa1_module, net.sa2_module, net.sa3_module):
        m.k = k                                          # hard-coded 32 at model.py:210-212

    rec = {"idx": [], "edges": []}
    orig_cc, orig_knn, orig_rad = ref_model.consecutive_cluster, ref_model.knn, ref_model.radius

    def cc(src):
        r = orig_cc(src); rec["idx"].append(r[1]); return r

    def knn_(*a, **kw):
        r = orig_knn(*a, **kw); rec["edges"].append(r); return r

    def rad_(*a, **kw):
        r = orig_rad(*a, **kw); rec["edges"].append(r); return r

    ref_model.consecutive_cluster, ref_model.knn, ref_model.radius = cc, knn_, rad_
    feats = {}
    hooks = []
    for mod_name in ("sa1_module", "sa2_module", "sa3_module", "sa4_module",
                     "fp4_module", "fp3_module", "fp2_module", "fp1_module"):
        hooks.append(getattr(net, mod_name).register_forward_hook(
            lambda m, i, o, n=mod_name: feats.__setitem__(n + ".out", o[0])))
    for mod_name in ("sa1_module", "sa2_module", "sa3_module"):
        hooks.append(getattr(net, mod_name).conv.register_forward_hook(
            lambda m, i, o, n=mod_name: feats.__setitem__(n + ".conv", o)))
    try:
        batch = synth.collate(voxels)
        d = _Data()
        d.pos, d.batch = batch["pos"].clone(), batch["batch"]
        d.reflectance, d.sf = batch["reflectance"].clone(), batch["sf"]
        with torch.no_grad():
            logits = net(d)
    finally:
        ref_model.consecutive_cluster, ref_model.knn, ref_model.radius = orig_cc, orig_knn, orig_rad
        for h in hooks:
            h.remove()

    out = {}
    for kname in ("pos", "batch", "reflectance", "sf", "local_shift", "ptr"):
        _store(out, "in." + kname, batch[kname])
    out["meta"] = np.array([C, k, wseed, len(voxels)], dtype=np.int64)
    if "recipe" in opts:   # inputs too large to store: the generator call per voxel (kind, side, points, seed, reflectance)
        out["in.recipe"] = np.array([[1 if r[0] == "uniform" else 2, r[1], r[2], r[3], r[4]] for r in opts["recipe"]], dtype=np.float64)
    for l in range(3):
        _store(out, f"idx{l+1}", rec["idx"][l])
        _store(out, f"edge{l+1}.q", rec["edges"][l][0])
        _store(out, f"edge{l+1}.c", rec["edges"][l][1])
    # the k=2 searches inside knn_interpolate go through oracle.ops.knn directly (fp4..fp1)
    _store(out, "stem", d.x)
    for n, t in feats.items():
        _store(out, n, t)
    _store(out, "logits", logits.reshape(-1), full="logits" in opts.get("full", ()))
    _store(out, "probs", torch.sigmoid(logits.reshape(-1)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    lv = logits.reshape(-1)
    info = {
        "file": name + ".npz", "C": C, "k": k, "weight_seed": wseed, "voxels": [int(v["pos"].shape[0]) for v in voxels],
        "M": [int(rec["idx"][l].numel()) for l in range(3)], "E": [int(rec["edges"][l].shape[1]) for l in range(3)],
        "logit_mean": float(lv.mean()), "logit_std": float(lv.std()) if lv.numel() > 1 else 0.0,
        "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "bytes": os.path.getsize(path),
    }
    print(json.dumps(info))
    return info


def main():
    only = set(sys.argv[1:])
    man_path = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(man_path)) if (only and os.path.exists(man_path)) else {}
    for name, voxels, C, k, wseed, *rest in _cases():
        if only and name not in only:
            continue
        manifest[name] = run_case(name, voxels, C, k, wseed, *rest)
    manifest["_generator"] = {"torch": torch.__version__, "numpy": np.__version__,
                              "reference": "harryjfowen/PointsToWood @ 2025-09-12 (/root/reference)",
                              "note": "reference src/model.py + src/pointnet.py imported over oracle/stubs"}
    json.dump(manifest, open(man_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
