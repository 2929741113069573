#!/usr/bin/env python3
"""Generate the HOST-SIDE golden fixtures in tests/golden/host/ FROM THE REFERENCE ITSELF.

Companion of make_golden.py (which pins ``Net.forward``): this script imports the reference's unmodified
``src/predicter.py``, ``src/io.py`` and ``predict.py`` from ``/root/reference/pointstowood`` in the authoring container
- over ``oracle/stubs`` for the packages that are not installed here (``torch_geometric``, ``torch_scatter``,
``pykdtree`` -> scipy cKDTree, ``numba`` -> identity decorators) - runs the rows of SURVEY.md 8a-14..17 and 8f-1/3 and
records inputs and outputs:

  feed.npz           TestingDataset.__getitem__ (predicter.py:78-94), incl. the NaN quirks
  sampler.json       BalancedBatchSampler (:23-63) under fixed np.random seeds
  load_model.json    load_model (:97-105): per-key checksums of the model after loading a ``module.``-prefixed checkpoint
  vote.npz           PointCloudClassifier.compute_labels (:112-127), k = 64 / 32, any_wood = 1 / 0.5
  collect.npz        collect_predictions (:129-142) on lattice coordinates (exact in fp32 and fp64)
  ply_*.ply, ply.npz write_ply bytes / read_ply arrays (io.py:11-83)
  columns.json       preprocess_point_cloud_data (predict.py:36-52)
  segmentation.npz   one CPU SemanticSegmentation run (:148-236) over a 4-voxel directory
  voxelise.npz       Voxelise.write_voxels / preprocess (preprocessing.py:79-131) on a small plot, with and without reflectance

The reference's source never enters this repo; only these data vectors do.

    python tests/golden/make_golden_host.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "host")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.insert(0, "/root/reference/pointstowood")

import src.io as ref_io  # noqa: E402
import src.predicter as ref  # noqa: E402
from pointstowood_amd import synthetic_weights as weights  # noqa: E402


def raw_voxels(seed=9):
    """Voxel tensors [n, 6] like the ones preprocess() writes (x, y, z, reflectance, extra, n_z) at plot coordinates."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, n in enumerate((700, 300, 1100, 500)):
        out.append(torch.cat([torch.rand(n, 3, generator=g) * 2 + torch.tensor([40.0 * i, 7.0, 100.0]),
                              torch.rand(n, 1, generator=g) * 2 - 1, torch.rand(n, 2, generator=g)], 1))
    return out


def gen_feed():
    vox = raw_voxels()
    nan_refl = vox[1].clone()
    nan_refl[[5, 77], 3] = float("nan")            # NaN reflectance: those rows are dropped, shift / sf unaffected
    nan_xyz = vox[3].clone()
    nan_xyz[10, 1] = float("nan")                  # NaN coordinate: the mean is NaN -> every row is NaN after the shift
    cases = [vox[0], nan_refl, vox[2], nan_xyz]
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for i, v in enumerate(cases):
            torch.save(v, os.path.join(d, f"voxel_{i}.pt"))
        ds = ref.TestingDataset(voxels=d, max_pts=16384, device="cpu")
        assert len(ds) == len(cases)
        for i, v in enumerate(cases):
            item = ds[i]
            out[f"raw{i}"] = v.numpy()
            out[f"pos{i}"] = item.pos.numpy()
            out[f"reflectance{i}"] = item.reflectance.numpy()
            out[f"local_shift{i}"] = item.local_shift.numpy()
            out[f"sf{i}"] = np.asarray(item.sf.numpy())
        # collation of the two clean voxels through the (stub) PyG loader path: shapes of sf / local_shift
        from torch_geometric.data import Batch
        b = Batch.from_data_list([ds[0], ds[2]])
        for k in ("pos", "reflectance", "local_shift", "sf", "batch", "ptr"):
            out["batch." + k] = getattr(b, k).numpy()
    np.savez_compressed(os.path.join(OUT, "feed.npz"), **out)


def gen_sampler():
    cases = []
    g = np.random.default_rng(3)
    for n, bs, seed in ((4, 2, 0), (9, 4, 1), (16, 8, 2), (7, 2, 3), (5, 3, 4), (12, 6, 5)):
        lengths = [int(v) for v in g.integers(128, 16384, n)]
        with tempfile.TemporaryDirectory() as d:
            for i, ln in enumerate(lengths):
                torch.save(torch.zeros(ln, 4), os.path.join(d, f"voxel_{i:03d}.pt"))
            ds = ref.TestingDataset(voxels=d, max_pts=16384, device="cpu")
            sampler = ref.BalancedBatchSampler(ds, bs)
            np.random.seed(seed)
            first = [[int(i) for i in b] for b in sampler]
            second = [[int(i) for i in b] for b in sampler]          # the global RNG has advanced: a different epoch
            cases.append({"lengths": lengths, "batch_size": bs, "seed": seed, "len": len(sampler), "epoch0": first,
                          "epoch1": second})
    json.dump(cases, open(os.path.join(OUT, "sampler.json"), "w"), indent=1)


def _checksums(sd):
    return {k: [float(v.double().sum()), float(v.double().abs().sum()), list(v.shape)] for k, v in sd.items()}


def gen_load_model():
    sd = weights.synth_state_dict(1, 32, seed=4)
    ck = {"module." + k: v for k, v in sd.items()}
    dropped = "fp2_module.NN.1.2.running_mean"
    del ck["module." + dropped]                      # missing key: strict=False keeps the model's own value
    ck["module.not_in_the_model.weight"] = torch.ones(3)   # unexpected key: ignored
    torch.manual_seed(0)
    net = ref.Net(num_classes=1).eval()
    before = net.state_dict()[dropped].clone()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.pth")
        torch.save({"model_state_dict": ck}, path)
        ref.load_model(path, net, "cpu")
        try:
            torch.save({"weights": sd}, path)
            ref.load_model(path, net, "cpu")
            err = None
        except KeyError as e:                          # predicter.py:162-165 catches exactly this
            err = repr(e)
    after = net.state_dict()
    assert torch.equal(after[dropped], before)
    json.dump({"weight_seed": 4, "dropped": dropped, "dropped_value": _checksums({dropped: before})[dropped],
               "extra": "not_in_the_model.weight", "missing_top_level_key_error": err, "loaded": _checksums(after)},
              open(os.path.join(OUT, "load_model.json"), "w"))


def gen_vote():
    g = np.random.default_rng(11)
    out = {}
    for k in (64, 32):
        n = 160
        nb = np.zeros((n, k, 5))
        nb[:, :, :3] = g.random((n, k, 3))
        nb[:, :, 4] = g.random((n, k))
        nb[:, :, 3] = (nb[:, :, 4] >= 0.5).astype(np.float64)
        nb[:40, :, 4] = np.round(nb[:40, :, 4] * 4) / 4            # ties in the median and in the vote sums
        nb[:40, :, 3] = (nb[:40, :, 4] >= 0.5).astype(np.float64)
        nb[40:60, :, 3] = 0.0                                        # unanimous neighbourhoods
        nb[60:80, :, 3] = 1.0
        out[f"nbr{k}"] = nb
        for any_wood in (1, 0.5, 0.9):
            labels = ref.PointCloudClassifier.compute_labels(nb, np.zeros((n, 2)), any_wood)
            out[f"labels{k}_{any_wood}"] = labels
    np.savez_compressed(os.path.join(OUT, "vote.npz"), **out)


def gen_collect():
    """collect_predictions end to end.  Coordinates on a 1/1024 m lattice with a little jitter-free structure: exact in
    fp32 and fp64, so the (third-party, here scipy-shimmed) neighbour search and an fp32 GPU search see the same
    distances; rows whose k-th and (k+1)-th neighbour are equidistant are flagged (their neighbour SET is not unique)."""
    from scipy.spatial import cKDTree
    g = np.random.default_rng(5)
    m, n = 6000, 1500
    cls_xyz = g.integers(0, 4096, (m, 3)) / 1024.0
    prob = g.random(m)
    classification = np.concatenate([cls_xyz, (prob >= 0.5)[:, None].astype(np.float64), prob[:, None]], 1)
    orig_xyz = g.integers(0, 4096, (n, 3)) / 1024.0
    out = {"classification": classification, "original_xyz": orig_xyz}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)                                   # collect_predictions looks for ./nbrs.npy
        try:
            for any_wood in (1, 0.5):
                k = 64 if any_wood == 1 else 32
                df = pd.DataFrame(np.concatenate([orig_xyz, g.random((n, 1))], 1), columns=["x", "y", "z", "reflectance"])
                res = ref.PointCloudClassifier(0.5, any_wood).collect_predictions(classification, df)
                out[f"label_{any_wood}"] = res["label"].to_numpy()
                out[f"pwood_{any_wood}"] = res["pwood"].to_numpy()
                dist, _ = cKDTree(cls_xyz).query(orig_xyz, k=k + 1)
                out[f"unique_{any_wood}"] = dist[:, k] > dist[:, k - 1]
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "collect.npz"), **out)


def gen_ply():
    g = np.random.default_rng(2)
    n = 257
    cols = {"x": g.random(n) * 1e5, "y": g.random(n) * 1e6, "z": g.random(n) * 50,
            "reflectance": (g.random(n) * 40 - 30).astype(np.float32),
            "red": g.integers(0, 256, n), "green": g.integers(0, 256, n), "blue": g.integers(0, 256, n),
            "n_z": g.random(n), "label": (g.random(n) > 0.5).astype(np.float64), "pwood": g.random(n)}
    df = pd.DataFrame(cols)
    path = os.path.join(OUT, "ply_written_by_reference.ply")
    ref_io.write_ply(path, df.copy())
    back = ref_io.read_ply(path)
    out = {"in." + k: np.asarray(v) for k, v in cols.items()}
    out["columns"] = np.array(list(back.columns))
    for c in back.columns:
        out["read." + c] = back[c].to_numpy()
    plain = pd.DataFrame({k: cols[k] for k in ("x", "y", "z", "reflectance")})
    ref_io.write_ply(os.path.join(OUT, "ply_written_by_reference_xyzr.ply"), plain.copy(), comments=["plot 7", "tile 3"])
    # an ascii and a float32 / uchar binary file as another tool would write them, read by the reference's reader
    apath = os.path.join(OUT, "ply_ascii.ply")
    with open(apath, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\n"
                "property float scalar_Reflectance\nend_header\n")
        for i in range(5):
            f.write(f"{i * 0.5} {i * 0.25 - 1} {10 - i} {-3.5 * i}\n")
    a = ref_io.read_ply(apath)
    out["ascii.columns"] = np.array(list(a.columns))
    for c in a.columns:
        out["ascii." + c] = a[c].to_numpy()
    bpath = os.path.join(OUT, "ply_binary_mixed.ply")
    rec = np.zeros(9, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<u2"), ("red", "u1"), ("dev", "<f8")])
    rec["x"], rec["y"], rec["z"] = g.random(9), g.random(9), g.random(9)
    rec["intensity"], rec["red"], rec["dev"] = g.integers(0, 60000, 9), g.integers(0, 255, 9), g.random(9)
    with open(bpath, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 9\nproperty float x\nproperty float y\nproperty float z\n"
                b"property uint16 intensity\nproperty uchar red\nproperty double dev\nend_header\n")
        rec.tofile(f)
    b = ref_io.read_ply(bpath)
    out["mixed.columns"] = np.array(list(b.columns))
    for c in b.columns:
        out["mixed." + c] = b[c].to_numpy()
    np.savez_compressed(os.path.join(OUT, "ply.npz"), **out)


def gen_columns():
    import predict as ref_predict   # the reference's predict.py (module level is import-safe; __main__ is guarded)
    cases = []
    for cols in (["X", "Y", "Z", "scalar_Reflectance", "dev", "label", "pwood"],
                 ["x", "y", "z", "red", "green", "blue", "Intensity", "n_z"],
                 ["x", "y", "z"],
                 ["x", "y", "z", "gps_time", "refl", "pleaf"]):
        df = pd.DataFrame(np.arange(3 * len(cols), dtype=np.float64).reshape(3, len(cols)), columns=cols)
        out, headers, has = ref_predict.preprocess_point_cloud_data(df)
        cases.append({"in": cols, "out": list(out.columns), "headers": list(headers), "reflectance": bool(has),
                      "values": out.to_numpy().tolist()})
    json.dump(cases, open(os.path.join(OUT, "columns.json"), "w"), indent=1)


def gen_segmentation():
    """SemanticSegmentation (predicter.py:148-236) on CPU over a directory of four voxels, batch_size 2: the whole row
    chain dataset -> sampler -> loader -> Net.forward (the reference's own, over the operator stubs) -> sigmoid /
    threshold / un-shift -> kd-tree + vote -> PLY."""
    vox = raw_voxels(seed=21)
    sd = weights.synth_state_dict(1, 32, seed=0)
    g = np.random.default_rng(8)
    allpts = torch.cat(vox, 0).numpy().astype(np.float64)
    sel = np.sort(g.choice(allpts.shape[0], 1200, replace=False))
    pc = pd.DataFrame(allpts[sel][:, [0, 1, 2, 3, 5]], columns=["x", "y", "z", "reflectance", "n_z"])
    captured = {}
    orig_collect = ref.PointCloudClassifier.collect_predictions

    def spy(self, classification, original):
        captured["classification"] = classification.copy()
        return orig_collect(self, classification, original)
    ref.PointCloudClassifier.collect_predictions = spy
    cwd = os.getcwd()
    try:
        with tempfile.TemporaryDirectory() as d:
            os.chdir(d)
            os.makedirs(os.path.join(d, "model"))
            torch.save({"model_state_dict": {"module." + k: v for k, v in sd.items()}}, os.path.join(d, "model", "m.pth"))
            vdir = os.path.join(d, "voxels")
            os.makedirs(vdir)
            for i, v in enumerate(vox):
                torch.save(v, os.path.join(vdir, f"voxel_{i}.pt"))
            args = types.SimpleNamespace(wdir=d, model="m.pth", vxfile=vdir, max_pts=16384, batch_size=2, is_wood=0.5,
                                         any_wood=1, verbose=False, pc=pc.copy(), headers=["reflectance"],
                                         odir=os.path.join(d, "plot_ours.ply"))
            np.random.seed(6)
            torch.set_num_threads(min(os.cpu_count() or 1, 16))
            args = ref.SemanticSegmentation(args)
            written = ref_io.read_ply(args.odir)
            ply_bytes = open(args.odir, "rb").read()
    finally:
        os.chdir(cwd)
        ref.PointCloudClassifier.collect_predictions = orig_collect
    out = {f"voxel{i}": v.numpy() for i, v in enumerate(vox)}
    out.update({"pc": pc.to_numpy(), "pc_columns": np.array(list(pc.columns)), "np_seed": np.array(6), "weight_seed": np.array(0),
                "classification": captured["classification"], "label": args.pc["label"].to_numpy(),
                "pwood": args.pc["pwood"].to_numpy(), "written_columns": np.array(list(written.columns)),
                "written_sha256": np.array(hashlib.sha256(ply_bytes).hexdigest())})
    for c in written.columns:
        out["written." + c] = written[c].to_numpy()
    np.savez_compressed(os.path.join(OUT, "segmentation.npz"), **out)


def gen_voxelise():
    """Voxelise.write_voxels (preprocessing.py:79-127) + preprocess (:129-131).  The reference hard-codes device='cuda'
    (:44-45,84,87); for this run ``Tensor.to`` / ``torch.arange`` are wrapped so that 'cuda' means the CPU - the
    arithmetic is the same torch code either way.  max_pts is above every voxel's size so no random capping happens
    (that part draws from the global torch RNG and is not comparable)."""
    import glob
    import src.preprocessing as ref_pp
    from tests.test_host_cpu import _plot
    real_to, real_arange = torch.Tensor.to, torch.arange

    def to(self, *a, **kw):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        if isinstance(kw.get("device"), str) and kw["device"].startswith("cuda"):
            kw["device"] = "cpu"
        return real_to(self, *a, **kw)

    def arange(*a, **kw):
        if isinstance(kw.get("device"), str) and kw["device"].startswith("cuda"):
            kw["device"] = "cpu"
        return real_arange(*a, **kw)
    out = {}
    torch.Tensor.to, torch.arange = to, arange
    ref_pp.torch.arange = arange
    try:
        for tag, refl in (("refl", True), ("norefl", False), ("refl_ties", True), ("has_nz", True)):
            pc = _plot(n=9000 if tag in ("refl", "norefl") else 4000, seed=3 if refl else 4, refl=refl)
            if tag == "refl":    # continuous reflectance: no ties, so the (unstable) sort of preprocessing.py:22 has one answer
                pc[:, 3] = torch.randperm(pc.shape[0], generator=torch.Generator().manual_seed(12)).float() / pc.shape[0] * 40 - 30
            cols = ["x", "y", "z", "reflectance", "dev"][: pc.shape[1]]
            if tag == "has_nz":   # e.g. a *_ours.ply fed back in: gpu_ground is skipped, the last column is taken for n_z
                pc[:, 3] = torch.randperm(pc.shape[0], generator=torch.Generator().manual_seed(13)).float() / pc.shape[0] * 40 - 30
                pc[:, 4] = pc[:, 2] - pc[:, 2].min()
                cols = ["x", "y", "z", "reflectance", "n_z"]
            df = pd.DataFrame(pc.numpy().astype(np.float64), columns=cols)
            with tempfile.TemporaryDirectory() as d:
                args = types.SimpleNamespace(pc=df, vxfile=d, min_pts=64, max_pts=100000, resolution=0.01, grid_size=[2.0, 4.0])
                ref_pp.preprocess(args)
                files = sorted(glob.glob(os.path.join(d, "voxel_*.pt")), key=lambda f: int(os.path.basename(f)[6:-3]))
                vox = [torch.load(f).numpy() for f in files]
            out[f"{tag}.pc"] = pc.numpy()
            out[f"{tag}.n_z"] = args.pc["n_z"].to_numpy()
            out[f"{tag}.count"] = np.array(len(vox))
            out[f"{tag}.sizes"] = np.array([v.shape[0] for v in vox])
            out[f"{tag}.voxels"] = np.concatenate(vox, 0)
    finally:
        torch.Tensor.to, torch.arange = real_to, real_arange
        ref_pp.torch.arange = real_arange
    np.savez_compressed(os.path.join(OUT, "voxelise.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    for name, fn in (("feed", gen_feed), ("sampler", gen_sampler), ("load_model", gen_load_model), ("vote", gen_vote),
                     ("collect", gen_collect), ("ply", gen_ply), ("columns", gen_columns), ("segmentation", gen_segmentation), ("voxelise", gen_voxelise)):
        if only and name not in only:
            continue
        fn()
        print("wrote", name, flush=True)
    info = {"torch": torch.__version__, "numpy": np.__version__, "pandas": pd.__version__,
            "reference": "harryjfowen/PointsToWood @ 2025-09-12 (/root/reference)",
            "note": "reference src/predicter.py, src/io.py, predict.py imported over oracle/stubs (pykdtree -> scipy cKDTree, "
                    "numba.jit -> identity, torch_geometric.data/loader -> oracle/stubs/torch_geometric)",
            "files": {f: hashlib.sha256(open(os.path.join(OUT, f), "rb").read()).hexdigest()
                      for f in sorted(os.listdir(OUT)) if f != "manifest.json"}}
    json.dump(info, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
