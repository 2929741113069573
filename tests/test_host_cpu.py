"""CPU tests of the host-side mirror of the reference interface: Data/Batch collation, dataset feed, sampler,
checkpoint loading, C-ABI symbol export, sharding plan, and the gloo world-size-2 gather."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import host as ohost
from pointstowood_amd import synthetic_voxels as synth, synthetic_weights as weights
from pointstowood_amd import Batch, Data, DataLoader, Net, checkpoint_layout
from pointstowood_amd import _lib
from pointstowood_amd.dist import gather_logits, partition_batches
from pointstowood_amd.predicter import BalancedBatchSampler, VoxelDataset, load_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _raw_voxels():
    g = torch.Generator().manual_seed(0)
    out = []
    for n in (300, 128, 999, 512, 640):
        pc = torch.rand(n, 5, generator=g) * 3 + 10
        out.append(pc)
    out[2][5, 3] = float("nan")      # NaN reflectance: the row is dropped
    out[2][7, 3] = float("nan")
    out[4][9, 1] = float("nan")      # NaN coordinate: poisons mean/sf, so the reference drops the WHOLE voxel
    return out


def test_dataset_feed_matches_oracle_restatement():
    raw = _raw_voxels()
    ds = VoxelDataset(raw)
    for i, pc in enumerate(raw):
        d, ref = ds[i], ohost.feed(pc)
        for k in ("pos", "reflectance", "local_shift", "sf"):
            a, b = getattr(d, k), ref[k]
            assert a.shape == b.shape and torch.allclose(a, b, rtol=0, atol=0, equal_nan=True), k
    assert ds[2].pos.shape[0] == 997  # two NaN rows dropped AFTER shift/sf were computed
    assert ds[4].pos.shape[0] == 0    # same quirk as the reference (predicter.py:81-90)


def test_collation_contract():
    ds = VoxelDataset(_raw_voxels())
    b = Batch.from_data_list([ds[0], ds[1], ds[3]])
    assert b.pos.shape == (940, 3) and b.reflectance.shape == (940,)
    assert b.sf.shape == (3,) and b.local_shift.shape == (9,)
    assert b.batch.dtype == torch.int64 and b.batch.tolist() == [0] * 300 + [1] * 128 + [2] * 512
    assert b.ptr.tolist() == [0, 300, 428, 940]
    ref = synth.collate([ohost.feed(_raw_voxels()[i]) for i in (0, 1, 3)])
    for k in ("pos", "reflectance", "local_shift", "sf", "batch", "ptr"):
        assert torch.equal(getattr(b, k), ref[k]), k
    b.x = torch.zeros(3)          # attribute assignment (model.py:228)
    assert b.to("cpu") is b
    loader = DataLoader(ds, batch_sampler=[[0, 1], [2, 3, 4]])
    sizes = [bb.pos.shape[0] for bb in loader]
    assert sizes == [428, 997 + 512 + 0]


def test_consume_unshift_matches_oracle():
    ds = VoxelDataset(_raw_voxels())
    b = Batch.from_data_list([ds[0], ds[2]])
    logits = torch.linspace(-3, 3, b.pos.shape[0])
    logits[3] = float("nan")
    ref = ohost.consume(logits, b.pos, b.batch, b.local_shift, 0.5)

    class M:
        def __call__(self, data):
            return logits
    from pointstowood_amd.predicter import classify_batch
    got = classify_batch(M(), b, 0.5, "cpu")
    assert np.allclose(got, ref, rtol=0, atol=0)


def test_samplers():
    ds = VoxelDataset(_raw_voxels())
    s = BalancedBatchSampler(ds, 2)
    batches = list(s)
    assert sorted(i for b in batches for i in b) == [0, 1, 2, 3, 4] and len(batches) == len(s) == 3
    assert list(BalancedBatchSampler(ds, 2)) == batches                 # deterministic
    ref = list(BalancedBatchSampler(ds, 2, reference=True))
    assert all(len(b) == 2 for b in ref) and len(ref) == 2               # remainder dropped like the reference
    with pytest.raises(ValueError):
        list(BalancedBatchSampler(ds, 1, reference=True))


def test_checkpoint_layout_and_load_model(tmp_path):
    assert [(k, tuple(s)) for k, s, _ in checkpoint_layout(1, 32)] == [(k, tuple(s)) for k, s, _ in weights.key_table(1, 32)]
    sd = weights.synth_state_dict(1, 8, seed=2)
    path = tmp_path / "m.pth"
    torch.save({"model_state_dict": {"module." + k: v for k, v in sd.items()}}, path)
    net = load_model(str(path), Net(1, C=8), "cpu")
    got = net.state_dict()
    assert list(got.keys()) == list(sd.keys()) and all(torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(KeyError):
        torch.save({"wrong": {}}, path)
        load_model(str(path), Net(1, C=8), "cpu")


def test_forward_refuses_cpu_tensors():
    net = Net(1, C=8)
    v = synth.collate([synth.uniform_voxel(2.0, 256, 1)])
    d = Data(pos=v["pos"], batch=v["batch"], reflectance=v["reflectance"], sf=v["sf"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(d)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "p2w.h")).read()
    declared = set(re.findall(r"\b(p2w_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"p2w_stream_t", "p2w_epilogue"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    h = ctypes.CDLL(_lib.LIB_PATH) if os.path.exists(_lib.LIB_PATH) else _lib.lib()
    for name in declared:
        assert hasattr(h, name), name
    L = _lib.lib()
    assert L.p2w_version() >= 100 and L.p2w_strerror(-4) == b"p2w: workspace too small"


def _header_prototypes():
    """name -> (return type, [parameter types]) of every function include/p2w.h declares, comments and parameter names stripped."""
    hdr = open(os.path.join(ROOT, "include", "p2w.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"^\s*#.*$", " ", hdr, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(p2w_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", hdr, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        types = []
        if params and params != "void":
            for prm in params.split(","):
                prm = " ".join(prm.split())
                t = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", prm).strip() if not prm.endswith("*") else prm   # drop the parameter's name
                types.append(t.replace(" *", "*"))
        protos[name] = (ret.replace(" *", "*"), types)
    return protos


def _ctype_class(t):
    """C type of the header -> the ctypes class the binding must use."""
    t = t.replace("const ", "").strip()
    if t.endswith("*") or t == "p2w_stream_t":
        return ctypes.c_char_p if t == "char*" else ctypes.c_void_p
    return {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
            "double": ctypes.c_double, "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64}[t]


def test_binding_signatures_match_the_header_prototypes():
    """Every prototype of include/p2w.h against pointstowood_amd/_lib.SIGNATURES: arity, return type and the class of every
    parameter (pointer / int32 / int64 / size_t / float / double) - a drifted binding reads garbage instead of failing."""
    protos = _header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, (ret, params) in protos.items():
        res, args = _lib.SIGNATURES[name]
        want_ret = None if ret == "void" else _ctype_class(ret)
        assert res is want_ret or (want_ret is ctypes.c_void_p and res is ctypes.c_char_p), (name, ret, res)
        assert len(args) == len(params), (name, len(args), params)
        for i, (a, t) in enumerate(zip(args, params)):
            want = _ctype_class(t)
            ok = a is want or (want is ctypes.c_void_p and isinstance(a, type) and issubclass(a, ctypes._Pointer))
            assert ok, (name, i, t, a)
    # the epilogue structure, field by field
    hdr = open(os.path.join(ROOT, "include", "p2w.h")).read()
    body = re.sub(r"/\*.*?\*/", " ", re.search(r"typedef struct p2w_epilogue \{(.*?)\} p2w_epilogue;", hdr, flags=re.S).group(1), flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        base = decl.split(",")[0]
        ptr = "*" in base
        names = [n.strip().lstrip("*") for n in re.sub(r"^(const )?[a-z0-9_]+\s*\*?", "", decl, count=1).split(",")]
        fields += [(n, ctypes.c_void_p if ptr else ctypes.c_int32) for n in names]
    assert fields == [(n, t) for n, t in _lib.Epilogue._fields_], fields


def test_header_compiles_as_c_and_the_library_links(tmp_path):
    """include/p2w.h is a C header (no C++ leaks) and libp2w_gfx950.so is an ordinary shared library: a C program built with gcc
    calls the version / error-string / packed-dimension helpers and gets the documented status from an argument error - no Python,
    no torch, no GPU.  The library exports the p2w_* ABI and nothing else."""
    import shutil
    import subprocess
    _lib.lib()
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('''
#include <stdio.h>
#include <string.h>
#include "p2w.h"
int main(void) {
    int32_t np = 0, kp = 0;
    p2w_epilogue e;
    memset(&e, 0, sizeof e);
    if (p2w_version() < 500) return 1;
    if (strcmp(p2w_strerror(P2W_EWORKSPACE), "p2w: workspace too small") != 0) return 2;
    p2w_packed_dims(100, 70, &np, &kp);
    if (np != 256 || kp != 96) return 3;
    if (p2w_gemm_h2(7, (const void*)16, 32, (const void*)16, 1.0f, 4, 4, 4, &e, (float*)16, 4, 0, 0, 0, 0) != P2W_EINVAL) return 4;
    if (p2w_gemm_h2_sk_ws_bytes() == 0) return 5;
    printf("abi ok %d %zu\\n", (int)p2w_version(), sizeof(p2w_epilogue));
    return 0;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-x", "c", str(src), "-I", os.path.join(ROOT, "include"), "-L", libdir,
                        "-l:libp2w_gfx950.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("abi ok"), (r.returncode, r.stdout, r.stderr)
    assert int(r.stdout.split()[3]) == ctypes.sizeof(_lib.Epilogue)
    nm = shutil.which("nm")
    if nm:
        syms = [l.split()[-1] for l in subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout.splitlines() if l.strip()]
        assert syms and all(x.startswith("p2w_") for x in syms), [x for x in syms if not x.startswith("p2w_")][:5]


def test_partition_is_balanced_and_deterministic():
    costs = [16384, 128, 9000, 700, 16384, 5000, 12000, 300, 8000]
    plan = partition_batches(costs, 4)
    assert sorted(i for p in plan for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in plan]
    assert max(loads) <= 1.35 * (sum(costs) / 4) and plan == partition_batches(costs, 4)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts = [5, 9]
    mine = torch.arange(counts[rank], dtype=torch.float32) + 100 * rank
    ragged = gather_logits(mine, dist, counts=counts)
    equal = gather_logits(torch.full((4,), float(rank)), dist)
    q.put((rank, ragged.tolist(), equal.tolist()))
    dist.destroy_process_group()


def test_gather_logits_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(30) for p in ps]
    expect = list(map(float, range(5))) + [100.0 + i for i in range(9)]
    for _, ragged, equal in res:
        assert ragged == expect and equal == [0.0] * 4 + [1.0] * 4


def test_batch_cost_follows_the_flop_formula():
    """dist.batch_cost = SURVEY 8(d)'s MAC formula on occupancy-estimated level sizes: exact coefficients, right
    magnitude for the canonical voxel, and NOT proportional to the point count."""
    from pointstowood_amd.dist import batch_cost
    u16k = batch_cost(16384)
    assert abs(u16k / 139.63e9 - 1.0) < 0.08                       # U2-16k = 139.63 GMAC
    assert batch_cost(0) == 0.0 and batch_cost(2048) > 0
    per_point_small, per_point_big = batch_cost(2048) / 2048, u16k / 16384
    assert per_point_small > 1.5 * per_point_big                   # small voxels keep more level-2/3 points per input point
    assert batch_cost(16384, volume=64.0) > u16k                   # a sparse 4 m voxel: almost every point survives sampling


class _FakeModel:
    """Stands in for Net on the CPU: logits = a fixed function of the positions."""
    def __call__(self, data):
        return data.pos.sum(dim=1) * 0.5 - 0.1 * data.batch.to(torch.float32)


def _sharded_worker(rank, world, port, vdir, q):
    import torch.distributed as dist
    from pointstowood_amd.predicter import VoxelDataset, classify_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = VoxelDataset(vdir)
    sampler = BalancedBatchSampler(ds, 2)           # deterministic (seeded) default mode: identical on every rank
    loads = []
    orig = ds.__class__.__getitem__

    def spy(self, i):
        loads.append(i)
        return orig(self, i)
    ds.__class__.__getitem__ = spy
    out = classify_sharded(_FakeModel(), ds, [list(b) for b in sampler], 0.5, "cpu", dist)
    q.put((rank, out.tolist(), sorted(loads)))
    dist.destroy_process_group()


def test_classify_sharded_gloo_world2(tmp_path):
    """Two ranks classify disjoint shares of the voxel batches (one voxel has NaN-reflectance rows that only its owner
    discovers), nobody loads a voxel it does not own, and every rank ends up with the single-process result."""
    import torch.multiprocessing as mp
    from pointstowood_amd.predicter import VoxelDataset, classify
    g = torch.Generator().manual_seed(5)
    for i, n in enumerate((300, 900, 150, 700, 500, 1100, 250)):
        pc = torch.cat([torch.rand(n, 3, generator=g) * 2 + 10 * i, torch.rand(n, 1, generator=g)], 1)
        if i == 3:
            pc[[1, 50, 51], 3] = float("nan")
        torch.save(pc, tmp_path / f"voxel_{i}.pt")
    ds = VoxelDataset(str(tmp_path))
    single = classify(_FakeModel(), DataLoader(ds, batch_sampler=BalancedBatchSampler(ds, 2), num_workers=0), 0.5, "cpu")
    assert single.shape == (3900 - 3, 5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 1000
    ps = [ctx.Process(target=_sharded_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(30) for p in ps]
    loads = [set(r[2]) for r in res]
    assert loads[0] and loads[1] and not (loads[0] & loads[1]) and loads[0] | loads[1] == set(range(7))
    for _, out, _ in res:
        assert np.array_equal(np.asarray(out, dtype=single.dtype), single)


def _cli_plan_worker(rank, world, port, vdir, q):
    import torch.distributed as dist
    from pointstowood_amd.predicter import VoxelDataset, classify_sharded, plan_batches
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = VoxelDataset(vdir)
    out = classify_sharded(_FakeModel(), ds, plan_batches(ds, 8, False, max_points=1500, max_voxels=3), 0.5, "cpu", dist)
    q.put((rank, out.tolist()))
    dist.destroy_process_group()


def test_cli_plan_gives_the_same_rows_on_one_and_on_two_ranks(tmp_path):
    """predict.py --voxels plans its forwards ONE way (predicter.plan_batches: point-budget batches) whether it runs as one process
    (classify_voxels) or sharded (classify_sharded): the rows are equal, row for row (logits depend on a batch's composition -
    here through _FakeModel's batch term, in the product through the batch-global grid origin)."""
    import torch.multiprocessing as mp
    from pointstowood_amd.predicter import VoxelDataset, classify_voxels
    g = torch.Generator().manual_seed(6)
    for i, n in enumerate((300, 900, 150, 700, 500, 1100, 250, 130)):
        torch.save(torch.cat([torch.rand(n, 3, generator=g) * 2 + 10 * i, torch.rand(n, 1, generator=g)], 1), tmp_path / f"voxel_{i}.pt")
    single = classify_voxels(_FakeModel(), VoxelDataset(str(tmp_path)), 0.5, "cpu", batch_size=8, max_points=1500, max_voxels=3)
    assert single.shape == (4030, 5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 11) % 1000
    ps = [ctx.Process(target=_cli_plan_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(30) for p in ps]
    for _, out in res:
        assert np.array_equal(np.asarray(out, dtype=single.dtype), single)


def test_serial_host_ops_restores_the_thread_count():
    from pointstowood_amd.data import serial_host_ops
    n = torch.get_num_threads()
    with serial_host_ops():
        assert torch.get_num_threads() == 1
    assert torch.get_num_threads() == n


def test_predict_cli_keeps_the_reference_flag_surface():
    import importlib.util
    spec = importlib.util.spec_from_file_location("p2w_predict", os.path.join(ROOT, "predict.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.build_parser().parse_args([])
    # names, types and defaults of pointstowood/predict.py:61-74
    assert (a.point_cloud, a.odir, a.batch_size, a.num_procs, a.resolution) == ([], ".", 8, -1, 0.01)
    assert (a.grid_size, a.min_pts, a.max_pts, a.model) == ([2.0, 4.0], 128, 16384, "model.pth")
    assert (a.is_wood, a.any_wood, a.output_fmt, a.verbose) == (0.5, 1, "ply", False)
    b = mod.build_parser().parse_args(["-p", "a.ply", "b.ply", "--is-wood", "0.7", "--grid_size", "2.0"])
    assert b.point_cloud == ["a.ply", "b.ply"] and b.is_wood == 0.7 and b.grid_size == [2.0]


def _plot(n=60000, seed=0, refl=True):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([13.0, 9.0]) + torch.tensor([100.0, -40.0])
    ground = 0.05 * xy[:, :1] + 0.3 * torch.sin(xy[:, 1:] / 3)
    z = ground + torch.rand(n, 1, generator=g) ** 2 * 11.0
    cols = [xy, z, (torch.randint(0, 200, (n, 1), generator=g).float() / 7 - 9) if refl else torch.zeros(n, 1),
            torch.rand(n, 1, generator=g)]
    return torch.cat(cols, 1)


def test_voxeliser_refuses_host_tensors():
    """One path per stage: the product's voxeliser runs its grid step on the HIP library and refuses points on the host."""
    from pointstowood_amd import preprocessing as PP
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PP.voxelise(_plot(n=2000), (2.0,), min_pts=64)


@pytest.mark.parametrize("refl", [True, False])
def test_voxeliser_matches_reference_restatement(refl, tensor_backend):
    from oracle import preprocess as OP
    from pointstowood_amd import preprocessing as PP
    pc = _plot(refl=refl)
    ref, nz_ref = OP.voxelise(pc, (2.0, 4.0), min_pts=64, max_pts=100000)
    got, nz = PP.voxelise(pc, (2.0, 4.0), min_pts=64, max_pts=100000, mode="compat")
    assert torch.equal(nz, nz_ref)
    assert len(got) == len(ref) and len(got) > 20
    for a, b in zip(got, ref):
        assert a.shape == b.shape and torch.equal(a, b)       # same voxels, same order, same rows
    # capping: sizes and membership (the sampled subset depends on the RNG stream)
    capped, _ = PP.voxelise(pc, (4.0,), min_pts=64, max_pts=300, mode="compat", generator=torch.Generator().manual_seed(1))
    full, _ = PP.voxelise(pc, (4.0,), min_pts=64, max_pts=100000, mode="compat")
    assert len(capped) == len(full) and all(c.shape[0] == min(f.shape[0], 300) for c, f in zip(capped, full))
    for c, f in zip(capped, full):
        fs = {tuple(r.tolist()) for r in f}
        assert all(tuple(r.tolist()) in fs for r in c[:20])
    xyz, _ = PP.voxelise(pc, (2.0,), min_pts=64, mode="xyz")
    assert sum(v.shape[0] for v in xyz) >= sum(v.shape[0] for v in PP.voxelise(pc, (2.0,), min_pts=64)[0])


def test_collate_device_equals_dataset_path():
    from pointstowood_amd.predicter import collate_device
    raw = [r for i, r in enumerate(_raw_voxels()) if i in (0, 1, 3)]      # NaN-free voxels
    ds = VoxelDataset(raw)
    ref = Batch.from_data_list([ds[i] for i in range(3)])
    got = collate_device(raw)
    assert torch.equal(got.batch, ref.batch) and torch.equal(got.ptr, ref.ptr) and torch.equal(got.reflectance, ref.reflectance)
    assert (got.local_shift - ref.local_shift).abs().max() <= 1e-5 and (got.sf - ref.sf).abs().max() <= 1e-5
    assert (got.pos - ref.pos).abs().max() <= 1e-5


def test_point_budget_sampler_covers_everything_within_budget():
    from pointstowood_amd.predicter import PointBudgetSampler
    g = torch.Generator().manual_seed(0)
    lengths = [int(v) for v in (torch.rand(500, generator=g) ** 3 * 16000 + 128)]
    s = PointBudgetSampler(lengths, max_points=40000, max_voxels=32)
    batches = list(s)
    assert sorted(i for b in batches for i in b) == list(range(500)) and len(batches) == len(s)
    assert all(sum(lengths[i] for i in b) <= 40000 and len(b) <= 32 for b in batches)
    fill = sum(lengths) / (len(batches) * 40000)
    assert fill > 0.8 and list(PointBudgetSampler(lengths, 40000, 32)) == batches


# ---- PLY I/O and CLI column handling (next row 8f-3) -------------------------------------------------------------
def test_ply_round_trip_binary_ascii_and_big_endian(tmp_path):
    from pointstowood_amd import io as pio
    g = np.random.default_rng(0)
    n = 257
    cols = {"x": g.random(n) * 100, "y": g.random(n), "z": g.random(n), "scalar_Reflectance": g.random(n).astype(np.float32),
            "red": g.integers(0, 255, n), "green": g.integers(0, 255, n), "blue": g.integers(0, 255, n)}
    p = tmp_path / "a.ply"
    pio.write_ply(str(p), cols, comments=["unit test"])
    back = pio.read_ply(str(p))
    assert list(back) == ["x", "y", "z", "red", "green", "blue", "scalar_Reflectance"]        # io.py:63-78 column order
    assert back["x"].dtype == np.float64 and back["red"].dtype == np.int32
    for k in cols:
        assert np.array_equal(back[k], np.asarray(cols[k]).astype(back[k].dtype))
    # a column that does not convert to float64 is dropped silently, as the reference's writer does (io.py:72-78)
    q = tmp_path / "text.ply"
    pio.write_ply(str(q), {**cols, "species": np.array(["oak"] * n), "flag": np.ones(n, dtype=bool)})
    assert list(pio.read_ply(str(q))) == ["x", "y", "z", "red", "green", "blue", "scalar_Reflectance", "flag"]
    # ascii and big-endian inputs
    header = "ply\nformat {} 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty double z\nproperty uchar intensity\nend_header\n"
    a = tmp_path / "b.ply"
    a.write_text(header.format("ascii") + "1 2 3 7\n4 5 6 8\n-1.5 0 2.25 255\n")
    got = pio.read_ply(str(a))
    assert got["x"].dtype == np.float32 and got["intensity"].dtype == np.uint8
    assert np.array_equal(got["z"], [3, 6, 2.25]) and np.array_equal(got["intensity"], [7, 8, 255])
    rec = np.zeros(3, dtype=[("x", ">f4"), ("y", ">f4"), ("z", ">f8"), ("intensity", "u1")])
    rec["x"], rec["z"], rec["intensity"] = [1, 4, -1.5], [3, 6, 2.25], [7, 8, 255]
    b = tmp_path / "c.ply"
    b.write_bytes(header.format("binary_big_endian").encode() + rec.tobytes())
    got = pio.read_ply(str(b))
    assert np.array_equal(got["x"], np.float32([1, 4, -1.5])) and np.array_equal(got["z"], [3, 6, 2.25])
    m = tmp_path / "mesh.ply"
    m.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n0\n3 0 0 0\n")
    with pytest.raises(ValueError, match="mesh"):
        pio.read_ply(str(m))
    t = tmp_path / "trunc.ply"
    t.write_bytes(header.format("binary_little_endian").encode() + b"\0" * 10)
    with pytest.raises(ValueError, match="truncated"):
        pio.read_ply(str(t))


def test_prepare_columns_follows_the_reference_cli():
    """predict.py:36-52: lower-case, drop label/pwood/pleaf, strip scalar_, refl|intensity -> reflectance at column 3."""
    from pointstowood_amd import io as pio
    n = 4
    z = np.zeros(n)
    cols, headers, had = pio.prepare_columns({"X": z, "Y": z, "Z": z, "red": z, "scalar_Intensity": z + 2, "label": z, "pwood": z})
    assert list(cols) == ["x", "y", "z", "reflectance", "red"] and headers == ["red", "reflectance"] and had
    assert np.array_equal(cols["reflectance"], z + 2)
    cols, headers, had = pio.prepare_columns({"x": z, "y": z, "z": z, "scalar_dev": z})
    assert list(cols) == ["x", "y", "z", "reflectance", "dev"] and headers == ["dev"] and not had
    assert np.array_equal(cols["reflectance"], z)
    cols, headers, had = pio.prepare_columns({"x": z, "y": z, "z": z, "refl": z + 1})
    assert list(cols) == ["x", "y", "z", "reflectance"] and headers == ["reflectance"] and had


def test_backproject_oracle_known_answers():
    """compute_labels (predicter.py:112-127) on hand-made neighbourhoods."""
    from oracle import backproject as OB
    nb = np.zeros((3, 4, 5))
    nb[0, :, 3], nb[0, :, 4] = [1, 1, 0, 0], [0.9, 0.8, 0.1, 0.2]      # wood votes 1.7 vs 0.3
    nb[1, :, 3], nb[1, :, 4] = [1, 0, 0, 0], [0.6, 0.4, 0.4, 0.4]      # 0.6 vs 1.2
    nb[2, :, 3], nb[2, :, 4] = [1, 0, 1, 0], [0.5, 0.5, 0.5, 0.5]      # tie -> first maximum = class 0
    lab = OB.compute_labels(nb, 1)
    assert lab[:, 0].tolist() == [1, 0, 0]
    assert np.allclose(lab[:, 1], [0.5, 0.4, 0.5])                       # medians: (0.2+0.8)/2, 0.4, 0.5
    assert OB.compute_labels(nb, 0.5)[:, 0].tolist() == [1, 1, 1]       # any neighbour predicted wood
    nb[1, :, 3] = 0
    assert OB.compute_labels(nb, 0.5)[:, 0].tolist() == [1, 0, 1]


def _rows_worker(rank, world, port, q):
    import torch.distributed as dist
    from pointstowood_amd.dist import gather_rows
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = [3, 0][rank] if world == 2 else 1
    rows = torch.arange(n * 4, dtype=torch.float32).reshape(n, 4) + 100 * rank
    out = gather_rows(rows, dist)
    empty = gather_rows(torch.zeros((0, 2)), dist)
    q.put((rank, out.tolist(), list(empty.shape)))
    dist.destroy_process_group()


def test_gather_rows_gloo_world2_ragged_and_empty():
    """The plot pipeline's two exchanges (classified points, per-point results) use this: lengths unknown in advance,
    a rank may contribute nothing."""
    import torch.multiprocessing as mp
    from pointstowood_amd.dist import slice_for_rank
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    ps = [ctx.Process(target=_rows_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(30) for p in ps]
    expect = torch.arange(12, dtype=torch.float32).reshape(3, 4).tolist()
    for _, out, eshape in res:
        assert out == expect and eshape == [0, 2]
    cover = [slice_for_rank(10, r, 4) for r in range(4)]
    assert cover == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [slice_for_rank(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]


def test_product_code_never_imports_the_oracle():
    """The oracle is the checker: the package, predict.py and the non-baseline part of bench.py must not touch it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pointstowood_amd")):
        for f in files:
            if f.endswith(".py"):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f
    assert not pat.search(open(os.path.join(ROOT, "predict.py")).read())
    src = open(os.path.join(ROOT, "bench.py")).read()
    hits = [m.start() for m in pat.finditer(src)]
    a, b = src.index("def cpu_baseline"), src.index("def main")
    assert hits and all(a < h < b for h in hits)          # only inside cpu_baseline()


def test_hot_kernels_stay_off_the_register_cliff():
    """The 256x256 GEMM kernel lives at 256 VGPRs with a small, epilogue-only scratch area; compiling extra code paths
    into it has silently doubled its scratch and halved its speed before.  Pin the resource usage of the hot kernels."""
    import shutil
    import subprocess
    from pointstowood_amd import build as B
    hipcc = B._hipcc()
    if shutil.which(hipcc) is None and not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    usage = {}
    for src in ("p2w_feat.hip", "p2w_feat_h1.hip", "p2w_geom.hip"):
        r = subprocess.run([hipcc, *B.FLAGS, "-c", os.path.join(B.CSRC, src), "-o", os.devnull,
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        name = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                usage[name] = {}
            for key in ("VGPRs Spill", "ScratchSize [bytes/lane]", "VGPRs", "LDS Size [bytes/block]"):
                m = re.search(re.escape(key) + r": (\d+)", line)
                if m and name:
                    usage[name].setdefault(key, int(m.group(1)))

    def one(substr):
        hits = [v for k, v in usage.items() if substr in k]
        assert len(hits) == 1, (substr, [k for k in usage if substr in k])
        return hits[0]
    for prec in (0, 1, 2):   # f16x3 (p2w_feat.hip), fp16 / bf16 (p2w_feat_h1.hip)
        for tile in ("Li2ELi4ELi4ELi2E", "Li2ELi2ELi2ELi2E"):   # the persistent GEMM, 256 x 256 and 128 x 128 tiles
            for dotk in ("Lb0E", "Lb1E"):                        # the plain kernel and its row-dot instantiation (the head)
                g = one(f"gemm_hp_kernelILi{prec}E{tile}{dotk}")
                assert g["ScratchSize [bytes/lane]"] <= 16, g      # 8 today: one value parked across the tile loop, not in the slab loop
        for k in (f"sa_conv16p_kernelILi{prec}ELi256ELi2ELi32E", f"sa_conv16p_kernelILi{prec}ELi128ELi2ELi32E",
                  f"sa_conv16p_kernelILi{prec}ELi256ELi2ELi8E", f"sa_conv16p_kernelILi{prec}ELi128ELi2ELi8E"):
            assert one(k)["VGPRs Spill"] == 0 and one(k)["LDS Size [bytes/block]"] <= 120 * 1024
    for k, v in usage.items():
        if "slab_search_kernel" in k or k.startswith("_Z10knn_kernel") or k.startswith("_Z11ball_kernel"):
            assert v["VGPRs Spill"] == 0, (k, v)


def test_pick_chunk_properties():
    from pointstowood_amd.engine import pick_chunk
    rounds = lambda rows, t: -(-(-(-rows // 256) * t) // 256)
    for m, budget, t in ((123046, 65536, 2), (81683, 32768, 4), (17506, 16384, 8), (131072, 61680, 2), (123046, 52428, 2),
                         (5, 65536, 2), (1000000, 65536, 2), (300, 100, 1)):
        c = pick_chunk(m, budget, t)
        assert c % 256 == 0 and c >= 256
        b = max(256, budget // 256 * 256)
        assert c <= max(b * 5 // 4, -(-m // 256) * 256)
        full, rem = divmod(m, c)
        cost = full * rounds(c, t) + (rounds(rem, t) if rem else 0)
        plain_full, plain_rem = divmod(m, b)
        assert cost <= plain_full * rounds(b, t) + (rounds(plain_rem, t) if plain_rem else 0)   # never worse than the plain budget
    assert pick_chunk(131072, 61680, 2) == 65536                                               # FP1: two full chunks, no 8192-row tail


def test_concurrent_lazy_builds_serialise(tmp_path):
    """Every rank of a torch.distributed.run launch may find the library stale at once (``*.so`` is not in git): exactly
    one of them must build, under the lock, and the others must come back with the fresh library."""
    import subprocess
    import sys
    script = f"""
import os, sys, time
sys.path.insert(0, {ROOT!r})
from pointstowood_amd import build as B
B.HERE = {str(tmp_path)!r}
B.LIB = os.path.join(B.HERE, "lib.so")
def fake(objdir, verbose):
    time.sleep(1.0)
    with open(os.path.join(B.HERE, "builds.log"), "a") as f:
        f.write(str(os.getpid()) + "\\n")
    with open(B.LIB + ".tmp" + str(os.getpid()), "w") as f:
        f.write("x" * 100000)
    os.replace(B.LIB + ".tmp" + str(os.getpid()), B.LIB)
    with open(B.LIB + ".srchash", "w") as f:
        f.write(B.source_hash())
    return B.LIB
B._build_locked = fake
print(B.build())
assert not B._stale() and os.path.getsize(B.LIB) == 100000
"""
    procs = [subprocess.Popen([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(4)]
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
    assert len(open(tmp_path / "builds.log").read().split()) == 1       # one builder, three waiters



# ---- multi-GPU readiness at world sizes 4 and 8 (gloo, CPU): partition + the plot pipeline's two exchanges -----------------
class _FakeStreamModel:
    """Net's streaming interface on the CPU: logits = a fixed function of positions and the voxel's own shift."""
    def stream(self, batches):
        for d in batches:
            yield (d.pos.sum(dim=1) + 0.01 * d.local_shift.view(-1, 3)[d.batch].sum(dim=1)) * 0.7 - 0.2


def _cpu_collect(cls_xyz, cls_pred, cls_prob, query_xyz, any_wood=1.0, k=1):
    """Stand-in for backproject.collect_predictions (HIP) in the CPU tests: nearest classified point's prediction."""
    if cls_xyz.shape[0] == 0 or query_xyz.shape[0] == 0:
        z = torch.zeros(query_xyz.shape[0])
        return z, z.clone()
    j = torch.cat([torch.cdist(q.to(torch.float64), cls_xyz.to(torch.float64)).argmin(dim=1) for q in query_xyz.split(4096)])
    return cls_pred[j].clone(), cls_prob[j].clone()


def _cpu_collect_checked(cls_xyz, cls_pred, cls_prob, query_xyz, any_wood=1.0, k=1):
    """Stand-in for backproject.collect_predictions_checked: the same nearest-point rule + the distance to that point (the "k-th"
    neighbour at k = 1), which the spatially sharded flow needs to prove a subset search exact.  Ties -> lowest index, like the product."""
    nq = query_xyz.shape[0]
    if cls_xyz.shape[0] == 0 or nq == 0:
        z = torch.zeros(nq)
        return z, z.clone(), torch.full((nq,), float("inf"), dtype=torch.float64)
    c64 = cls_xyz.to(torch.float64)
    parts = [torch.cdist(q.to(torch.float64), c64).min(dim=1) for q in query_xyz.split(4096)]     # (chunk by chunk: never the full matrix)
    dk, j = torch.cat([p.values for p in parts]), torch.cat([p.indices for p in parts])
    return cls_pred[j].clone(), cls_prob[j].clone(), dk


def _cpu_stand_ins():
    """In a spawned worker: the CPU stand-ins of the two GPU stages around the host-side logic under test."""
    from oracle import preprocess as OP
    from pointstowood_amd import pipeline, preprocessing
    pipeline.collect_predictions = _cpu_collect
    pipeline.collect_predictions_checked = _cpu_collect_checked
    preprocessing.backend = OP.TensorBackend
    return pipeline


def _plot_worker(rank, world, port, max_points, shard, q):
    import torch.distributed as dist
    pipeline = _cpu_stand_ins()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    calls = []
    orig_gather = dist.all_gather

    def counting_gather(out, t, *a, **kw):
        calls.append(int(t.numel()))
        return orig_gather(out, t, *a, **kw)
    dist.all_gather = counting_gather
    pc = _plot(n=12000, seed=3)
    stats = {}
    n_z, label, pwood = pipeline.segment_plot(pc, _FakeStreamModel(), (4.0,), min_pts=64, max_pts=100000, max_points=max_points,
                                              generator=torch.Generator().manual_seed(0), stats=stats, dist=dist, shard=shard, halo=0.5)
    dist.all_gather = orig_gather
    q.put((rank, label.tolist(), pwood.tolist(), float(n_z.double().sum()), stats.get("voxels"), len(calls), stats.get("backproject_tiers")))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,max_points,shard", [(4, 3000, "spatial"), (8, 100000, "spatial"), (4, 3000, "slices"), (8, 3000, "spatial")])
def test_segment_plot_sharded_gloo_world4_and_8(world, max_points, shard, monkeypatch, tensor_backend):
    """The sharded plot flow (pipeline.segment_plot) at world sizes 4 and 8, including ranks that get NO batch (world 8 with one
    large-budget batch: seven idle ranks) and ranks whose share is one small batch: every rank must end with the single-process
    result, point for point.  shard="spatial" (default): LPT-partitioned voxel batches -> ONE all-gather of the float32
    probabilities -> back-projection owned by x-slabs against the voxels within a halo (0.5 m here, so that the wider tiers are
    exercised: a query whose nearest classified point is farther than its distance to the covered range's end is repeated against
    a 4 x wider set) -> ONE all-gather of the results: exactly two collectives.  shard="slices": round 5's flow."""
    import torch.multiprocessing as mp
    from pointstowood_amd import pipeline
    from pointstowood_amd.dist import batch_cost, partition_batches
    from pointstowood_amd.predicter import PointBudgetSampler
    from pointstowood_amd.preprocessing import voxelise
    monkeypatch.setattr(pipeline, "collect_predictions", _cpu_collect)
    monkeypatch.setattr(pipeline, "collect_predictions_checked", _cpu_collect_checked)
    pc = _plot(n=12000, seed=3)
    n_z, label, pwood = pipeline.segment_plot(pc, _FakeStreamModel(), (4.0,), min_pts=64, max_pts=100000, max_points=max_points,
                                              generator=torch.Generator().manual_seed(0))
    vox, _ = voxelise(pc, (4.0,), 64, 100000, generator=torch.Generator().manual_seed(0))
    lengths = [int(v.shape[0]) for v in vox]
    batches = list(PointBudgetSampler(lengths, max_points, max(1, max_points // 1024)))   # segment_plot's default voxel cap
    plan = partition_batches([sum(batch_cost(lengths[i]) for i in b) for b in batches], world)
    assert sorted(i for p in plan for i in p) == list(range(len(batches)))
    if world == 8 and max_points == 100000:
        assert sum(1 for p in plan if not p) >= 1            # the case under test: idle ranks
    elif world == 4:
        assert len(batches) >= world and all(plan)            # every rank busy, shares of one or two batches
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + 13 * world + 7 * len(shard) + max_points // 1000) % 1000
    ps = [ctx.Process(target=_plot_worker, args=(r, world, port, max_points, shard, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(world))
    [p.join(60) for p in ps]
    for r in res:
        assert r[1] == label.tolist() and r[2] == pwood.tolist(), f"rank {r[0]} differs from the single-process result"
        assert r[3] == pytest.approx(float(n_z.double().sum()), rel=1e-6)
        assert r[5] == (2 if shard == "spatial" else 4)      # collectives of the data path (slices: two lengths + two blocks)
    if shard == "spatial":
        tiers = [r[6] for r in res if r[6]]
        assert tiers and any(len(t) > 1 for t in tiers), tiers      # the 0.5 m halo did leave queries for a wider tier somewhere


def test_x_slab_owners_are_balanced_contiguous_and_complete():
    """pipeline._x_slab_owners: every plot point has exactly one owner, slabs are contiguous in x (a slab's largest x is below the
    next one's smallest), sizes within a few histogram bins of n / world - also for clustered x (most points in 2 % of the range),
    constant x and an empty plot."""
    from pointstowood_amd.pipeline import _x_slab_owners
    g = torch.Generator().manual_seed(0)
    for name, x in (("uniform", torch.rand(50000, generator=g) * 100 - 50),
                    ("clustered", torch.cat([torch.rand(45000, generator=g) * 2 + 10, torch.rand(5000, generator=g) * 100 - 50])),
                    ("constant", torch.full((1000,), 3.0)), ("empty", torch.zeros(0))):
        for world in (1, 2, 8):
            owner, lists = _x_slab_owners(x, world)
            assert len(lists) == world and sum(o.numel() for o in lists) == x.numel(), (name, world)
            assert torch.equal(torch.sort(torch.cat(lists)).values, torch.arange(x.numel())), (name, world)
            mine = [x[o] for o in lists if o.numel()]
            assert all(float(a.max()) <= float(b.min()) for a, b in zip(mine[:-1], mine[1:])), (name, world)
            if name == "uniform":
                assert max(o.numel() for o in lists) <= 1.02 * x.numel() / world + 50, (name, world)
            if name == "clustered" and world == 8:     # a 4096-bin histogram still splits the dense 2 % of the range over several ranks
                assert max(o.numel() for o in lists) <= 0.2 * x.numel(), [o.numel() for o in lists]
            for o in lists:                            # every rank's list keeps the input order (the final scatter relies on the lists alone)
                assert torch.equal(o, torch.sort(o).values)


def _budget_worker(rank, world, port, q):
    import torch.distributed as dist
    pipeline = _cpu_stand_ins()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    pipeline.collect_predictions = _cpu_collect
    # every rank reads a DIFFERENT amount of free memory; rank 1's is the smallest and must set everybody's budget
    free = [40, 6, 17, 29][rank] * 1024 ** 3
    pipeline._free_bytes = lambda dev: free
    budget = pipeline.default_budget(10_000_000, world, free, dist)
    seen = []

    class _Sampler(list):
        def __init__(self, lengths, max_points, max_voxels):
            seen.append((max_points, max_voxels))
            super().__init__([[i] for i in range(len(lengths))])
    pipeline.PointBudgetSampler = _Sampler
    pc = _plot(n=12000, seed=3)
    n_z, label, pwood = pipeline.segment_plot(pc, _FakeStreamModel(), (4.0,), min_pts=64, max_pts=100000,
                                              generator=torch.Generator().manual_seed(0), dist=dist)
    q.put((rank, budget, seen[0], float(label.sum()), float(pwood.double().sum())))
    dist.destroy_process_group()


def test_default_budget_is_rank_invariant_gloo():
    """with different free memory per GPU every rank must still build the SAME batch list (ranks index into it);
    the cap is the all-reduced minimum."""
    import torch.multiprocessing as mp
    from pointstowood_amd import pipeline
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 1000
    ps = [ctx.Process(target=_budget_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    expect = int(0.4 * 6 * 1024 ** 3) // pipeline.BYTES_PER_POINT
    assert [r[1] for r in res] == [expect] * world                     # 10 M points / 20 = 500 k, capped by rank 1's 6 GiB
    assert len({r[2] for r in res}) == 1 and res[0][2][0] == min(262144, expect) > 65536
    assert len({r[3:] for r in res}) == 1                              # and the same plot result everywhere
    assert pipeline.default_budget(10_000_000, 1, None) == 2_000_000 and pipeline.default_budget(100, 1, None) == 262144


def _ragged_worker(rank, world, port, counts, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.arange(counts[rank], dtype=torch.float32) + 1000 * rank
    out = gather_logits(mine, dist, counts=counts)
    q.put((rank, out.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[5, 0, 9, 1], [3, 0, 0, 7, 1, 2, 0, 4]])
def test_gather_logits_gloo_world4_and_8_with_empty_and_tiny_ranks(counts):
    import torch.multiprocessing as mp
    world = len(counts)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + (os.getpid() + 7 * world) % 1000
    ps = [ctx.Process(target=_ragged_worker, args=(r, world, port, counts, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=300) for _ in range(world))
    [p.join(60) for p in ps]
    expect = [float(i + 1000 * r) for r, c in enumerate(counts) for i in range(c)]
    assert all(out == expect for _, out in res)


def test_auto_cell_and_default_forward_budget(monkeypatch, tensor_backend):
    """backproject.auto_cell follows the cloud's density (denser cloud -> smaller cells, clamped); segment_plot's default
    budget is a fifth of the classified points, clamped to [262144, 2097152] (so a small plot goes through in one forward)."""
    from pointstowood_amd import pipeline
    from pointstowood_amd.backproject import auto_cell
    g = torch.Generator().manual_seed(0)
    sparse = torch.rand(20000, 3, generator=g) * 50.0
    dense = torch.rand(20000, 3, generator=g) * 5.0
    assert 0.02 <= auto_cell(dense, 64) < auto_cell(sparse, 64) <= 2.0
    assert abs(auto_cell(sparse, 64) / auto_cell(dense, 64) - 10.0) < 0.5            # (k V / n)^(1/3): 10 x the extent
    assert auto_cell(torch.zeros(5, 3), 64) == 0.02 and auto_cell(sparse * 1e4, 64) == 2.0
    seen = {}

    class _Sampler(list):
        def __init__(self, lengths, max_points, max_voxels):
            seen["budget"] = (max_points, max_voxels)
            super().__init__([list(range(len(lengths)))])
    monkeypatch.setattr(pipeline, "PointBudgetSampler", _Sampler)
    monkeypatch.setattr(pipeline, "collect_predictions", _cpu_collect)
    pipeline.segment_plot(_plot(n=12000, seed=3), _FakeStreamModel(), (4.0,), min_pts=64, max_pts=100000,
                          generator=torch.Generator().manual_seed(0))
    assert seen["budget"] == (262144, 256)
