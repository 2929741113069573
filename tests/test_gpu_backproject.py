"""Back-projection (next row 8f-1): GPU grid kNN + vote against the CPU restatement of predicter.py:107-142."""
import numpy as np
import pytest
import torch

from oracle import backproject as OB

pytestmark = pytest.mark.gpu


def _scene(n_cls, n_q, seed, extent=(12.0, 9.0, 6.0), lattice=None):
    """Clustered 'plot': points on a few random planes and blobs.  With ``lattice`` every coordinate is a multiple of
    it (a dyadic step): fp32 and fp64 distances are then exact and equal, so both searches order candidates alike."""
    g = np.random.default_rng(seed)
    def pts(n):
        kind = g.integers(0, 3, n)
        p = g.random((n, 3)) * extent
        plane = kind == 0
        p[plane, 2] = 0.3 * np.sin(p[plane, 0]) + 1.0 + 0.01 * g.standard_normal(plane.sum())
        blob = kind == 1
        c = g.random((8, 3)) * extent
        p[blob] = c[g.integers(0, 8, blob.sum())] + 0.25 * g.standard_normal((blob.sum(), 3))
        return p
    cls, q = pts(n_cls), pts(n_q)
    if lattice:
        cls, q = np.round(cls / lattice) * lattice, np.round(q / lattice) * lattice
    prob = g.random(n_cls).astype(np.float32)
    pred = (prob >= 0.5).astype(np.float32)
    return cls.astype(np.float32), pred, prob, q.astype(np.float32)


def test_vote_matches_compute_labels_on_given_neighbours():
    from pointstowood_amd._lib import lib, ptr, stream
    g = np.random.default_rng(3)
    nc, n = 5000, 3000
    prob = g.random(nc).astype(np.float32)
    prob[g.integers(0, nc, 300)] = 0.5                       # repeated values in the median
    pred = (prob >= 0.5).astype(np.float32)
    for k, any_wood in ((64, 1.0), (32, 0.5), (32, 0.9)):
        nbr = g.integers(0, nc, (n, k)).astype(np.int32)
        cls = np.zeros((nc, 5)); cls[:, 3] = pred; cls[:, 4] = prob
        ref = OB.compute_labels(cls[nbr], any_wood)
        lab = torch.empty(n, dtype=torch.float32, device="cuda"); pw = torch.empty_like(lab)
        deg = torch.full((n,), k, dtype=torch.int32, device="cuda")
        t = lambda a: torch.from_numpy(a).cuda()
        nb, pd, pr = t(nbr), t(pred), t(prob)
        assert lib().p2w_vote(ptr(nb), ptr(deg), k, ptr(pd), ptr(pr), n, any_wood, ptr(lab), ptr(pw), stream()) == 0
        assert np.array_equal(lab.cpu().numpy(), ref[:, 0].astype(np.float32))
        assert np.array_equal(pw.cpu().numpy(), ref[:, 1].astype(np.float32))   # median: exact


@pytest.mark.parametrize("any_wood", [1.0, 0.5])
def test_collect_predictions_exact_on_lattice_coordinates(any_wood):
    """Coordinates on a 1/1024 m lattice: distances are exact in fp32 and fp64, index tie-breaks are the only freedom;
    labels / pwood must agree wherever the k-th and (k+1)-th distances differ."""
    from pointstowood_amd.backproject import collect_predictions
    cls, pred, prob, q = _scene(60000, 20000, 11, lattice=1.0 / 1024)
    k = 64 if any_wood == 1 else 32
    classification = np.concatenate([cls.astype(np.float64), pred[:, None], prob[:, None]], 1)
    ref = OB.collect_predictions(classification, q, any_wood)
    lab, pw = collect_predictions(*(torch.from_numpy(a).cuda() for a in (cls, pred, prob, q)), any_wood=any_wood)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(cls.astype(np.float64)).query(q.astype(np.float64), k=k + 1)
    untied = d[:, k - 1] < d[:, k]                                  # the k-set is unique
    assert untied.mean() > 0.9
    assert np.array_equal(lab.cpu().numpy()[untied], ref[untied, 0].astype(np.float32))
    assert np.array_equal(pw.cpu().numpy()[untied], ref[untied, 1].astype(np.float32))


def _twin_scene(offset, seed=5, n_cls=40000, n_q=30000):
    """Arbitrary float64 coordinates the way a plot produces them: every classified point occurs twice (the two voxel grid
    sizes), the copies a rounding apart (each is the float64 sum of a float32 position and a float32 shift,
    predicter.py:211), at a survey-grade offset.  float32 cannot tell the copies apart; float64 can."""
    cls, pred, prob, q = _scene(n_cls, n_q, seed)
    g = np.random.default_rng(seed + 100)
    base = cls.astype(np.float64) + np.asarray(offset, dtype=np.float64)
    twin = base + g.standard_normal(base.shape) * 2e-7
    cls64 = np.concatenate([base, twin])
    perm = g.permutation(len(cls64))
    cls64 = cls64[perm]
    prob2 = np.concatenate([prob, g.random(n_cls).astype(np.float32)])[perm]
    pred2 = (prob2 >= 0.5).astype(np.float32)
    q64 = q.astype(np.float64) + np.asarray(offset, dtype=np.float64) + g.standard_normal(q.shape) * 1e-4
    return cls64, pred2, prob2, q64


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (5.0e5, 6.2e6, 100.0)])
@pytest.mark.parametrize("any_wood", [1.0, 0.5])
def test_collect_predictions_general_coordinates_and_chunks(offset, any_wood):
    """Arbitrary float64 coordinates, near-coincident copies, a 5e5 / 6e6 m offset: the neighbour SETS are the float64
    KD-tree's (predicter.py:136-137) wherever the k-th and (k + 1)-th float64 distances differ - label and pwood EQUAL
    there, not 'mostly' - and chunking / cell size cannot matter."""
    from scipy.spatial import cKDTree
    from pointstowood_amd.backproject import collect_predictions, neighbours
    cls64, pred, prob, q64 = _twin_scene(offset)
    k = 64 if any_wood == 1 else 32
    classification = np.concatenate([cls64, pred[:, None].astype(np.float64), prob[:, None].astype(np.float64)], 1)
    ref = OB.collect_predictions(classification, q64, any_wood)
    d, ref_idx = cKDTree(cls64).query(q64, k=k + 1)
    untied = d[:, k - 1] < d[:, k]
    assert untied.mean() > 0.99
    t = [torch.from_numpy(a).cuda() for a in (cls64, pred, prob, q64)]
    lab, pw = collect_predictions(*t, any_wood=any_wood)
    lab2, pw2 = collect_predictions(*t, any_wood=any_wood, chunk=7000, cell=0.17)
    assert torch.equal(lab, lab2) and torch.equal(pw, pw2)
    assert np.array_equal(lab.cpu().numpy()[untied], ref[untied, 0].astype(np.float32))
    assert np.array_equal(pw.cpu().numpy()[untied], ref[untied, 1].astype(np.float32))
    # the index sets themselves, and their order (ascending float64 distance)
    got = np.full((len(q64), k), -1, dtype=np.int64)
    for rows, nbr, deg in neighbours(t[0], t[3], k, chunk=11000):
        assert bool((deg == k).all())
        got[rows.cpu().numpy()] = nbr.cpu().numpy()
    strict = untied & (np.diff(d[:, :k], axis=1) > 0).all(1)            # no ties inside the set either: the order is unique
    assert np.array_equal(np.sort(got[untied], 1), np.sort(ref_idx[untied, :k], 1))
    assert np.array_equal(got[strict], ref_idx[strict, :k])


def test_refine_survives_piles_of_coincident_points():
    """More candidates inside the fp32 result's bound than the refinement's list holds (300 exact copies of one point): the
    one-by-one extraction takes over; ties go to the lower index."""
    from pointstowood_amd.backproject import neighbours
    g = np.random.default_rng(1)
    cls = g.random((5000, 3)) * 4.0
    cls[1000:1300] = cls[1000]                                   # a pile of 300 coincident points
    q = np.concatenate([cls[1000][None] + 1e-3, g.random((200, 3)) * 4.0])
    t = [torch.from_numpy(a).cuda() for a in (cls, q)]
    got = np.zeros((len(q), 64), dtype=np.int64)
    for rows, nbr, deg in neighbours(t[0], t[1], 64):
        got[rows.cpu().numpy()] = nbr.cpu().numpy()
    d = ((q[:, None, :] - cls[None, :, :]) ** 2)
    d2 = (d[:, :, 0] + d[:, :, 1]) + d[:, :, 2]
    want = np.lexsort((np.broadcast_to(np.arange(len(cls)), d2.shape), d2), axis=1)[:, :64]
    assert np.array_equal(got, want)
    assert np.array_equal(got[0], np.arange(1000, 1064))


def test_fewer_classified_points_than_k():
    from pointstowood_amd.backproject import collect_predictions
    cls, pred, prob, q = _scene(20, 100, 2)
    lab, pw = collect_predictions(*(torch.from_numpy(a).cuda() for a in (cls, pred, prob, q)))
    assert np.allclose(pw.cpu().numpy(), np.median(prob))          # every query sees all 20
    want = 1.0 if prob[pred == 1].astype(np.float64).sum() > prob[pred == 0].astype(np.float64).sum() else 0.0
    assert bool((lab == want).all())


def test_predict_cli_point_cloud_end_to_end(tmp_path):
    """predict.py --point-cloud FILE.ply: PLY -> voxelise -> forward -> back-projection -> *_ours.ply, against the CPU
    restatements composed the same way (oracle preprocess / host / net / backproject)."""
    import importlib.util
    import os
    from oracle import host as ohost, net as onet, preprocess as OP
    from pointstowood_amd import synthetic_voxels as synth, synthetic_weights as weights
    from pointstowood_amd import io as pio
    from pointstowood_amd.predicter import PointBudgetSampler
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("p2w_predict", os.path.join(root, "predict.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.random.default_rng(4)
    n = 6000                                   # two dense clumps 30 m apart at a large easting: a handful of voxels
    c = np.array([[500000.0, 6200000.0, 100.0], [500030.0, 6200004.0, 103.0]])
    xyz = c[g.integers(0, 2, n)] + g.random((n, 3)) * [3.0, 3.0, 5.0]
    refl = g.random(n) * 40 - 30
    src = tmp_path / "plot.ply"
    pio.write_ply(str(src), {"x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2], "scalar_Reflectance": refl, "dev": g.random(n)})
    sd = weights.synth_state_dict(1, 32, seed=0)
    torch.save({"model_state_dict": sd}, tmp_path / "m.pth")
    out = mod.main(["--point-cloud", str(src), "--model", str(tmp_path / "m.pth"), "--min_pts", "64"])
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    assert out == [str(tmp_path / "plot_ours.ply")]
    got = pio.read_ply(out[0])
    assert list(got) == ["x", "y", "z", "reflectance", "dev", "n_z", "label", "pwood"]
    assert np.array_equal(got["x"], xyz[:, 0]) and np.array_equal(got["reflectance"], refl)   # inputs echoed untouched
    assert set(np.unique(got["label"])) <= {0.0, 1.0} and got["pwood"].min() >= 0 and got["pwood"].max() <= 1
    # CPU restatement of the same flow on the same plot-local fp32 coordinates
    local = np.concatenate([xyz - xyz.min(0), refl[:, None], got["dev"][:, None]], 1).astype(np.float32)
    vox, n_z = OP.voxelise(torch.from_numpy(local), (2.0, 4.0), min_pts=64, max_pts=16384)
    assert np.abs(got["n_z"] - n_z.numpy()).max() <= 1e-5
    lengths = [int(v.shape[0]) for v in vox]
    rows = []
    for batch in PointBudgetSampler(lengths, 262144, 256):   # segment_plot's default budget for a plot this small
        b = synth.collate([ohost.feed(vox[i]) for i in batch])
        logits = onet.forward(sd, b["pos"], b["batch"], b["reflectance"], b["sf"], k=32)
        rows.append(ohost.consume(logits, b["pos"], b["batch"], b["local_shift"], 0.5))
    classification = np.vstack(rows).astype(np.float64)
    ref = OB.collect_predictions(classification, local[:, :3].astype(np.float64), 1)
    assert (got["label"] == ref[:, 0]).mean() > 0.995        # decision-threshold and k-th-neighbour ties aside
    # pwood: the forward's 1e-4 carries straight through the median; where the two searches pick a different 64th
    # neighbour (coordinates differ in the last fp32 bit between the two flows) the median moves to an adjacent order
    # statistic instead - rare, and bounded by the local spread of the probabilities
    err = np.abs(got["pwood"] - ref[:, 1])
    assert (err < 2e-4).mean() > 0.98 and np.median(err) < 2e-5 and np.quantile(err, 0.999) < 0.1   # ~1/64 steps


def test_cell_start_table_and_search_through_it():
    """p2w_cell_starts = searchsorted of every cell id in the sorted keys; the search through the table returns exactly the
    neighbours the bisection path returns (table_cells = 0), for explicit and density-derived cell sizes."""
    import ctypes as C
    from pointstowood_amd import _lib
    from pointstowood_amd.backproject import neighbours, auto_cell
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    n_cells = 200_000
    keys = torch.randint(0, n_cells + 50, (300_000,), generator=g)   # (some keys beyond the table)
    keys[1000:41000] = 777                                            # one cell with 40 000 points
    keys = torch.sort(keys).values.cuda()
    table = torch.empty(n_cells + 1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(L.p2w_cell_starts_ws_bytes(n_cells)) + 256, dtype=torch.uint8, device="cuda")
    _lib.check(L.p2w_cell_starts(_lib.ptr(keys), keys.numel(), n_cells, _lib.ptr(table), _lib.ptr(ws), ws.numel(), _lib.stream()), "cell_starts")
    want = torch.searchsorted(keys, torch.arange(n_cells + 1, device="cuda"), right=False).to(torch.int32)
    assert torch.equal(table, want)
    _lib.check(L.p2w_cell_starts(None, 0, 10, _lib.ptr(table), _lib.ptr(ws), ws.numel(), _lib.stream()), "cell_starts")   # no keys
    assert int(table[:11].abs().sum()) == 0

    cls, pred, prob, q = _scene(60000, 20000, 9)
    c, qq = torch.from_numpy(cls).cuda(), torch.from_numpy(q).cuda()
    assert 0.02 <= auto_cell(c, 64) <= 2.0
    for cell in (None, 0.07, 0.4):
        a = [(r.clone(), n.clone(), d.clone()) for r, n, d in neighbours(c, qq, 64, cell)]
        b = [(r.clone(), n.clone(), d.clone()) for r, n, d in neighbours(c, qq, 64, cell, table_cells=0)]
        assert len(a) == len(b) and all(torch.equal(x, y) for ta, tb in zip(a, b) for x, y in zip(ta, tb))


class _ReplayDist:
    """torch.distributed for ONE rank of a pretended world on one GPU: all_gather records this rank's block per call index and
    plays back the blocks the other ranks recorded in an earlier pass (a rank whose peers are unknown gets its own block back)."""
    def __init__(self, rank, world, store):
        self.rank, self.world, self.store, self.calls = rank, world, store, 0

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return self.rank

    def all_gather(self, out, t, group=None):
        c, self.calls = self.calls, self.calls + 1
        slot = self.store.setdefault(c, {})
        slot[self.rank] = t.detach().clone()
        for r in range(self.world):
            out[r].copy_(slot.get(r, t))


@pytest.mark.parametrize("world,halo", [(4, 1.0), (3, 0.05)])
def test_spatially_sharded_backprojection_equals_one_process(world, halo):
    """pipeline.segment_plot(shard="spatial") with the REAL stages (voxeliser, Net, grid kNN + float64 refinement + vote), every
    rank's share run in turn on this GPU through a replaying stand-in for torch.distributed: pass 1 records every rank's
    probabilities, pass 2 back-projects each rank's x-slab against its halo's voxels (wider tiers where the k-th neighbour could lie
    outside), the assembled result equals the single-process labels and pwood BIT FOR BIT (k = 64 neighbour sets, ties by the
    global index)."""
    from pointstowood_amd import Net, pipeline
    from pointstowood_amd import synthetic_voxels as synth, synthetic_weights as weights
    dev = torch.device("cuda")
    net = Net(num_classes=1, C=8, k=16)
    net.load_state_dict(weights.synth_state_dict(1, 8, seed=0), strict=True)
    net = net.to(dev).eval()
    pc = synth.forest_plot(400_000, seed=4, side=22.0).to(dev)
    gen = lambda: torch.Generator(device=dev).manual_seed(0)
    kw = dict(grid_sizes=(2.0, 4.0), min_pts=128, max_pts=16384, max_points=65536, max_voxels=64)
    n_z1, label1, pwood1 = pipeline.segment_plot(pc, net, generator=gen(), **kw)
    # the voxel cells the flow selects candidates by hold every point of their voxel
    from pointstowood_amd.preprocessing import voxelise
    cells = []
    vox, _ = voxelise(pc, (2.0, 4.0), 128, 16384, generator=gen(), cells=cells)
    lo, hi = torch.cat([a for a, _ in cells]), torch.cat([b for _, b in cells])
    seg = torch.repeat_interleave(torch.arange(len(vox), device=dev), torch.tensor([v.shape[0] for v in vox], device=dev))
    allp = torch.cat([v[:, :3] for v in vox])
    assert lo.shape[0] == len(vox) and bool(((allp >= lo[seg]) & (allp <= hi[seg])).all())
    assert float((hi - lo).max()) < 4.02 and float((hi - lo).min()) > 2.0
    store = {}
    for r in range(world):      # pass 1: the ranks' probabilities
        pipeline.segment_plot(pc, net, generator=gen(), dist=_ReplayDist(r, world, store), halo=halo, **kw)
    parts, tiers = [], []
    for r in range(world):      # pass 2: with everybody's probabilities; the second exchange records the rank's results
        st = {}
        pipeline.segment_plot(pc, net, generator=gen(), dist=_ReplayDist(r, world, store), halo=halo, stats=st, **kw)
        tiers.append(st["backproject_tiers"])
    _, own = pipeline._x_slab_owners(pc[:, 0], world)
    assert sum(o.numel() for o in own) == pc.shape[0] and max(o.numel() for o in own) < 1.1 * pc.shape[0] / world     # balanced slabs
    label, pwood = torch.empty_like(label1), torch.empty_like(pwood1)
    for r in range(world):
        both = store[1][r][: 2 * own[r].numel()].view(-1, 2)
        label[own[r]], pwood[own[r]] = both[:, 0], both[:, 1]
    assert torch.equal(label, label1) and torch.equal(pwood, pwood1)
    if halo < 0.1:
        assert any(len(t) > 1 for t in tiers), tiers                        # a 5 cm halo cannot settle every query in the first tier
