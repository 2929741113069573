"""Host-side orchestration of the MI355X forward.

Replaces the body of ``Net.forward`` (reference ``pointstowood/src/model.py:226-245``) and
the modules it calls with two phases on one HIP stream:

* **geometry** - depends on positions only: record packing, the three grid sub-samplings
  (``SAModule.voxelsample``), ball query / kNN per SA level, the ``(p/sf)*sf`` level
  positions and the three k=2 searches of the feature-propagation levels.  Level sizes
  are data dependent and stay on the device; kernels are launched over upper bounds.
* **features** - after ONE small device-to-host copy of the three level sizes: stem,
  per level a hoisted layer-1 GEMM + the fused PointNetConv kernel + the 4-GEMM residual
  block, the global level, four interpolate+MLP levels and the head.

BatchNorm (eval) and the depthwise 1x1 convolutions are per-channel affines; they are
folded into the neighbouring GEMM's weights or epilogue when the checkpoint is packed
(``PackedWeights``), so a residual block is 4 GEMM launches instead of ~20 kernels.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
from dataclasses import dataclass, field, fields

import torch

from . import _lib
from ._lib import GEMM_RESIDUAL_H, PREC_F16X3, PREC_OF, SA_PACK8, SEARCH_BOX, SEARCH_Q_ROW_IN_W, SEARCH_X_INDEX_IN_W, Epilogue, check, lib, ptr

BN_EPS = 1e-5
SA_RES = (0.04, 0.08, 0.16)  # model.py:210-212


# --------------------------------------------------------------------------- weights
@dataclass
class Linear:
    """One packed GEMM: W [N_pad, K_pad] (k contiguous, zero padded) + epilogue vectors."""
    w: torch.Tensor
    N: int
    K: int
    bias: torch.Tensor | None = None
    sc0: torch.Tensor | None = None
    sh0: torch.Tensor | None = None
    sc1: torch.Tensor | None = None
    sh1: torch.Tensor | None = None
    relu0: int = 0
    relu1: int = 0
    relu2: int = 0
    relu_final: int = 0
    w16: torch.Tensor | None = None   # H weights of W * 2^e: [N_pad, 2 K_pad] fp16, blocks of 32 k as [hi | lo] (f16x3),
                                      # [N_pad, K_pad] fp16 / bf16 (single-plane precisions)
    wscale: float = 1.0               # 2^-e


def _bn_affine(sd, p):
    s = sd[p + ".weight"].double() / torch.sqrt(sd[p + ".running_var"].double() + BN_EPS)
    t = sd[p + ".bias"].double() - sd[p + ".running_mean"].double() * s
    return s, t


class PackedWeights:
    """Device-resident, kernel-ready form of the reference's 257-key state dict."""

    def __init__(self, sd, C_: int, num_classes: int, device, precision: str = "f16x3"):
        self.C, self.num_classes, self.device, self.precision = C_, num_classes, device, precision
        prec = PREC_OF.get(precision)   # None: fp32 MFMA (no H weights needed, K padded to 32)
        if C_ % 4 != 0:
            raise ValueError("C must be a multiple of 4 (16-byte feature rows)")
        sd = {k: v.detach().cpu() for k, v in sd.items()}
        f32 = lambda t: t.to(torch.float32).contiguous().to(device)
        self._f32 = f32

        def pack(W, **kw):
            W = W.double()
            N, K = W.shape
            Np, Kp = _lib.packed_dims(N, K, prec)
            Wp = torch.zeros((Np, Kp), dtype=torch.float64)
            Wp[:N, :K] = W
            vec = {k: (f32(v) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
            if prec is None:
                return Linear(w=f32(Wp), N=N, K=K, **vec)
            # 16-bit forms: scale by a power of two so the largest weight sits near 2^10 (f16x3: lo parts stay normal fp16)
            amax = float(Wp.abs().max())
            e = int(math.floor(math.log2(1024.0 / amax))) if amax > 0 else 0
            e = max(-24, min(24, e))
            Ws = Wp * (2.0 ** e)
            if prec == PREC_F16X3:
                hi = Ws.to(torch.float32).to(torch.float16)
                lo = (Ws - hi.double()).to(torch.float32).to(torch.float16)
                # H layout (include/p2w.h): per block of 32 k [hi(32) | lo(32)], one row per output channel
                w16 = torch.stack([hi.view(Np, Kp // 32, 32), lo.view(Np, Kp // 32, 32)], dim=2).reshape(Np, 2 * Kp)
                w16 = w16.contiguous().to(device)
            else:   # one plane, round to nearest
                w16 = Ws.to(torch.float32).to(torch.float16 if precision == "fp16" else torch.bfloat16).contiguous().to(device)
            return Linear(w=None, N=N, K=K, w16=w16, wscale=2.0 ** (-e), **vec)

        self.stem_w = f32(sd["stem_mlp.0.0.weight"])
        self.stem_b = f32(sd["stem_mlp.0.0.bias"])
        self.sa = []
        f_in = C_
        for l in (1, 2, 3):
            p = f"sa{l}_module"
            W1, b1 = sd[p + ".conv.local_nn.0.0.weight"], sd[p + ".conv.local_nn.0.0.bias"]
            C1 = W1.shape[0]
            W2, b2 = sd[p + ".conv.local_nn.1.0.weight"], sd[p + ".conv.local_nn.1.0.bias"]
            C2 = W2.shape[0]
            s2, t2 = _bn_affine(sd, p + ".conv.local_nn.1.2")
            _, C1p = _lib.packed_dims(C2, C1, prec)
            w1r4 = torch.zeros((4, C1p), dtype=torch.float32)
            w1r4[:, :C1] = W1[:, f_in:f_in + 4].t()
            lvl = {
                "F_in": f_in, "C1": C1, "C2": C2,
                "hoist": pack(W1[:, :f_in], bias=b1),               # P = x_src W1x^T + b1
                "w1r4": f32(w1r4),
                "w1r_bound": float(w1r4.abs().sum(dim=0).max()),   # |layer-1 correction| <= this (|normalised offset|, |reflectance| <= 1)
                "W2": pack(W2), "b2": f32(b2), "bn_s": f32(s2), "bn_t": f32(t2),
            }
            # InvertedResidualBlock (model.py:46-85) as 4 GEMMs
            r, F, E = p + ".residual_block", C2, 4 * C2
            se, te = _bn_affine(sd, r + ".expand.1")
            dws, dwt = self._dw_affine(sd, r + ".conv.0")
            lvl["g1"] = pack(sd[r + ".expand.0.weight"][:, :, 0].double() * se[:, None],
                             bias=se * sd[r + ".expand.0.bias"].double() + te, relu0=1, sc0=dws, sh0=dwt, relu1=1)
            sp, tp = _bn_affine(sd, r + ".conv.0.pointwise_bn")
            s1, t1 = _bn_affine(sd, r + ".conv.1")
            dws2, dwt2 = self._dw_affine(sd, r + ".conv.3")
            lvl["g2"] = pack(sd[r + ".conv.0.pointwise_conv.weight"][:, :, 0].double() * sp[:, None],
                             bias=sp * sd[r + ".conv.0.pointwise_conv.bias"].double() + tp, relu0=1,
                             sc0=s1, sh0=t1, relu1=1, sc1=dws2, sh1=dwt2, relu2=1)
            sp2, tp2 = _bn_affine(sd, r + ".conv.3.pointwise_bn")
            lvl["g3"] = pack(sd[r + ".conv.3.pointwise_conv.weight"][:, :, 0].double() * sp2[:, None],
                             bias=sp2 * sd[r + ".conv.3.pointwise_conv.bias"].double() + tp2, relu0=1)
            s4, t4 = _bn_affine(sd, r + ".conv.4")          # BN before the projection: fold into its input side
            spj, tpj = _bn_affine(sd, r + ".project.1")     # BN after it: fold into its output side
            Wp = sd[r + ".project.0.weight"][:, :, 0].double()
            lvl["g4"] = pack(spj[:, None] * Wp * s4[None, :],
                             bias=spj * (Wp @ t4 + sd[r + ".project.0.bias"].double()) + tpj, relu_final=1)
            self.sa.append(lvl)
            f_in = C2
        self.sa4 = self._mlp2(sd, "sa4_module.NN", pack)
        self.fp = {l: self._mlp2(sd, f"fp{l}_module.NN", pack) for l in (4, 3, 2, 1)}
        # layer 0 of an FP module (model.py:149-153) is linear up to its ReLU and knn_interpolate is a convex combination of coarse rows:
        #   relu(W [interp(y) | skip] + b) = relu(interp(W_i y) + W_s skip + b)
        # so W_i can be applied to the COARSE rows (a fifth of the fine ones at level 3 of the bench batch) and its output
        # interpolated - inside the epilogue of the GEMM over the skip columns (p2w_epilogue.interp), so the interpolated rows
        # never exist in HBM (EngineOptions.fp_hoist): (W_i without bias / activation, W_s with the bias, ReLU behind the sum)
        self.fp_split = {}
        for l in (4, 3, 2, 1):
            W0, b0 = sd[f"fp{l}_module.NN.0.0.weight"], sd[f"fp{l}_module.NN.0.0.bias"]
            Fc = self.sa4[1].N if l == 4 else self.fp[l + 1][1].N      # width of the coarse level's features
            self.fp_split[l] = (pack(W0[:, :Fc]), pack(W0[:, Fc:], bias=b0, relu_final=1))
        sn, tn = _bn_affine(sd, "norm")
        self.head1 = pack(sd["conv1.weight"][:, :, 0].double() * sn[:, None],
                          bias=sn * sd["conv1.bias"].double() + tn, relu0=1)
        self.head2_w = f32(sd["conv2.weight"][:, :, 0])
        self.head2_b = sd["conv2.bias"].to(torch.float32)
        self.head2 = pack(sd["conv2.weight"][:, :, 0], bias=sd["conv2.bias"]) if num_classes != 1 else None

    @staticmethod
    def _dw_affine(sd, p):
        """depthwise k=1 conv (per-channel w*x+b) followed by its BatchNorm = one affine."""
        s, t = _bn_affine(sd, p + ".depthwise_bn")
        w = sd[p + ".depthwise_conv.weight"][:, 0, 0].double()
        b = sd[p + ".depthwise_conv.bias"].double()
        return w * s, b * s + t

    @staticmethod
    def _mlp2(sd, p, pack):
        """MLP([a,b,c]) (model.py:198-202): Lin+ReLU, then Lin+ReLU+BN."""
        s, t = _bn_affine(sd, p + ".1.2")
        return (pack(sd[p + ".0.0.weight"], bias=sd[p + ".0.0.bias"], relu0=1),
                pack(sd[p + ".1.0.weight"], bias=sd[p + ".1.0.bias"], relu0=1, sc0=s, sh0=t))


# --------------------------------------------------------------------------- geometry
@dataclass
class Level:
    xyzr: torch.Tensor           # [n_bound, 4] records of this level
    ptr: torch.Tensor            # [B+1] int32 CSR (device)
    batch: torch.Tensor          # [n_bound] int32
    idx: torch.Tensor | None = None   # [n_bound] index into the previous level
    nbr: torch.Tensor | None = None   # [n_bound, k] neighbours in the previous level
    deg: torch.Tensor | None = None
    n: int = -1                  # exact size once known on the host


@dataclass
class Geometry:
    B: int
    N: int
    k: int
    sf: torch.Tensor
    levels: list = field(default_factory=list)   # 0..3
    fp_nbr: dict = field(default_factory=dict)   # fine level f -> (nbr [n,2], deg) into level f+1
    counts_dev: torch.Tensor | None = None
    counts_host: torch.Tensor | None = None
    done: object = None

    def tensors(self):
        out = [self.sf, self.counts_dev]
        for lv in self.levels:
            out += [t for t in (lv.xyzr, lv.ptr, lv.batch, lv.idx, lv.nbr, lv.deg) if t is not None]
        for nbr, deg in self.fp_nbr.values():
            out += [nbr, deg]
        out += list(getattr(self, "aux", []))
        return [t for t in out if t is not None]


def pick_chunk(m: int, budget: int, col_tiles: int, n_cu: int = 256, full_rounds: bool = False) -> int:
    """Rows per chunk for a row-chunked GEMM chain over ``m`` rows: a multiple of 256 within [0.6, 1.25] x ``budget`` (the
    row count that keeps the chain's intermediates cache-resident) that minimises the number of chip rounds of 256 x 256
    tiles over all chunks (``col_tiles`` column tiles per row tile) - a 1122-row tail chunk or a chunk that fills 1.1
    rounds costs as much as a full one.  Ties go to the larger chunk."""
    if m <= 0:
        return 256
    budget = max(256, budget // 256 * 256)
    if m <= budget * 5 // 4:
        # one chunk - unless its 256 x 256 tiles would fill the last of several chip rounds so badly that the GEMM falls
        # back to its small tile for ALL of them (the library takes the large tile from 78 % fill): then the whole rounds
        # go first as a chunk of their own and the remainder follows as a small one (level 3 of the bench batch:
        # 17506 rows x 8 column tiles = 2.16 rounds -> 16384 rows on the large tile + 1122 rows)
        per_round = max(256, 256 * n_cu // max(1, col_tiles) // 256 * 256)
        tiles = -(-m // 256) * col_tiles
        rounds_ = -(-tiles // n_cu)
        if full_rounds and m > per_round and tiles * 100 < rounds_ * n_cu * 78:
            return m // per_round * per_round
        return -(-m // 256) * 256
    best, best_cost = budget, None
    lo, hi = max(256, budget * 3 // 5 // 256 * 256), budget * 5 // 4 // 256 * 256
    for c in range(lo, hi + 1, 256):
        full, rem = divmod(m, c)
        rounds = lambda rows: -(-(-(-rows // 256) * col_tiles) // n_cu)
        cost = full * rounds(c) + (rounds(rem) if rem else 0)
        if best_cost is None or cost < best_cost or (cost == best_cost and c > best):
            best, best_cost = c, cost
    return best


@dataclass
class EngineOptions:
    """Behaviour switches of the forward.  They are keyword arguments of ``Net(...)`` / ``Engine(...)`` - the product reads
    no environment variables (nor does the C ABI behind it).  Every combination gives the same results; the defaults are
    the measured-fastest ones on MI355X, the others exist for A/B runs, tests and odd inputs."""
    sampler: str = "table"        # grid sub-sampling: "table" = direct cell table (no sort; per-batch fallback to the sort), "sort"
    search: str = "grid"          # neighbour searches: "grid" = cell-indexed (p2w_*_grid), "brute" = whole-voxel streaming kernels
    table_cells_per_point: float = 32.0   # the table sampler is taken while its table has at most this many entries per point
    search_index: bool = True     # grid searches look candidate runs up in the table sampler's cell -> position tables (else bisect)
    search_box: int = 0           # bit mask: grid searches bounded in x too (P2W_SEARCH_BOX: one run per grid row): 1 ball query,
                                  # 2 the k = 32 searches, 4 the interpolation searches (A/B: per-voxel rows are short)
    fp1_cell_order: bool = False  # H path: the level-0 features live in the sampler's cell order (stem, SA1's hoisted product, FP1, head), the
                                  # logits are scattered back, so that the last interpolation's coarse rows and SA1's P rows come from L2.
                                  # Bit-identical; measured: interp_concat 0.303 -> 0.289 ms, the rest of the step +-0 -> off (tools/opt_ab.py)
    fp_hints: bool = True         # seed the k = 2 interpolation searches from the sampler's ranks (p2w_knn_hint2)
    sa_pack: bool = True          # P2W_SA_PACK8 on the ball-query level (targets with <= 8 neighbours share an MFMA tile)
    chunk_pick: bool = True       # fill-aware row-chunk sizes (pick_chunk); False: the plain budget
    chunk_full_rounds: bool = True   # a residual-block level whose tiles fill its last chip round badly: whole rounds first, rest after
    res_chunk_rows: int = 131072  # rows (at 4F = 512) per residual-block / FP chunk; 0 = whole level (swept: tools/chunk_sweep.sh)
    res_streams: int = 1          # row-chunk chains (residual blocks, FP modules) in flight inside ONE feature phase.  2 fills the
                                  # round tails of a lone forward (-2 % of it) but adds nothing once Net.stream() keeps two feature
                                  # phases in flight - and every extra high-priority stream competes for the few hardware queues
                                  # (an idle third one cost the two-phase pipeline 5 %, measured): 1 by default
    feature_streams: int = 2      # Net.stream(): feature phases in flight (2: the kernels of batch i + 1 fill the round tails of batch
                                  # i's: -6 % of the bench step; 3: no further gain)
    geo_priority: int = 0         # HIP stream priorities of the two-stream pipeline (features are the critical path)
    feat_priority: int = -1
    res_priority: int = -1        # ... and of the second chunk-chain stream (part of the feature phase)
    gemm_flags: int = 0           # P2W_GEMM_* bits of include/p2w.h passed to every p2w_gemm_h2 call (A/B runs)
    fp_hoist: bool = True         # FP modules whose coarse level has at most fp_hoist_ratio of the fine level's rows: layer 0's interpolated half runs on
                                  # the coarse rows and its OUTPUT is interpolated inside the epilogue of the GEMM over the skip columns
                                  # (fewer MACs - level 3 of the bench batch has 17 506 coarse rows for 81 683 fine ones - and the
                                  # interpolated rows are neither written nor re-read); same function, logits move in the last bits
    fp_hoist_ratio: float = 0.5   # ... coarse rows / fine rows up to which a module takes that route (at 2/3 the gathers of the epilogue cost
                                  # what the saved interpolation kernel did: level 2 of the bench batch, tools/interp_epi_ab.py)
    range_guard: bool = True      # f16x3: every stem / PointNetConv / GEMM launch reports whether an output lay beyond +-6e4 (the hi plane
                                  # saturates at 65504) and whether any lay above 2^-5 (a tensor without one has its lo plane on fp16's
                                  # subnormal floor); a forward with such a layer is recomputed on the fp32 MFMA path (Engine.fallback, set
                                  # by Net) instead of returning silently degraded logits.  (fp16 / bf16 are throughput modes whose
                                  # error is reported, not bounded: no watch.)  The watch costs one host wait per forward for the finished
                                  # phase's report: Net.stream hides it behind the NEXT batch's feature phase only with feature_streams >= 2
                                  # (with one feature stream the GPU idles while the host looks at the report and enqueues the next batch);
                                  # "no value above 2^-5" is a SAMPLE (first tile of every persistent workgroup): it triggers the fp32
                                  # recomputation where one is available and never raises by itself
    gemm_stream_k: bool = True    # GEMMs run through p2w_gemm_h2_sk with a per-stream workspace: rows that do not fill a whole chip round
                                  # run as a stream-K tail where the library's cost model says it pays (then a level is ONE chunk:
                                  # chunk_full_rounds does not apply)
    sa_flags: int = 0             # P2W_SA_ITEM_* bits passed to p2w_sa_conv_h (A/B runs)
    overlap: bool = True          # ONE forward (model(data), the reference's call): the searches on a second stream beside the features
                                  # (each feature kernel waits for the event of the search it reads); False: strictly one stream
    search_priority: int = 0      # HIP priority of that stream (-1 = high)
    search_stagger: bool = False  # ... the later searches held back until the feature kernels they should run beside (A/B)
    early_first: bool = True      # ... and the size-independent feature kernels (stem, SA1's hoisted product) are enqueued BEFORE the searches
    table_prepared: bool = True   # table sampler through p2w_voxel_sample_table_prepared (5 launches per level; False: the 7-launch entry point)
    single_res_streams: int = 2   # row-chunk chains in flight inside a LONE forward (see res_streams: there the pipeline's second phase
                                  # fills the round tails; here nothing else does)

    def __post_init__(self):
        if self.sampler not in ("table", "sort"):
            raise ValueError("sampler must be 'table' or 'sort'")
        if self.search not in ("grid", "brute"):
            raise ValueError("search must be 'grid' or 'brute'")

    @classmethod
    def names(cls):
        return tuple(f.name for f in fields(cls))


class Engine:
    def __init__(self, weights: PackedWeights, k: int = 32, precision: str = "f16x3", options: EngineOptions | None = None, **kw):
        self.w = weights
        self.k = int(k)
        if precision not in ("f16x3", "fp32", "fp16", "bf16"):
            raise ValueError("precision must be 'f16x3' (split-fp16 MFMA, fp32-class accuracy: the parity default), 'fp32' "
                             "(fp32 MFMA), or 'fp16' / 'bf16' (one MFMA per product: the reference's autocast arithmetic, "
                             "does not meet the 1e-4 probability bar)")
        if getattr(weights, "precision", precision) != precision:
            raise ValueError(f"weights were packed for precision {weights.precision!r}, not {precision!r}")
        self.precision = precision
        self.prec = PREC_OF.get(precision)   # P2W_PREC_* of the H family, None for fp32
        if not 1 <= self.k <= 32:
            raise ValueError("k must be in 1..32: the fused PointNetConv maps a target's neighbour slots onto one 32-row MFMA "
                             "tile (P2W_MAX_K_CONV in include/p2w.h); the reference hard-codes k = 32 (model.py:210-212)")
        if options is not None and kw:
            raise TypeError("pass either options= or individual EngineOptions fields")
        self.options = options if options is not None else EngineOptions(**kw)
        for name in EngineOptions.names():      # the switches live on the engine (tools flip them between runs)
            setattr(self, name, getattr(self.options, name))
        self._table_scale = [1, 1, 1]   # per level: grows by 8 (up to TABLE_SCALE_MAX) after an overflow
        self._table_rest = [0, 0, 0]    # per level: batches the table sits out after overflowing at its largest scale
        self._table_probe = [False, False, False]   # per level: the table just handed out was smaller than the remembered factor (a probe)
        self._table_probe_rest = [0, 0, 0]          # per level: oversized batches that take the sort before the next probe
        self._ws_t = None
        self._range = None        # range watch of the feature phase being enqueued: (layer names, device floats)
        self.fallback = None      # callable(geo, keep) -> logits on an arithmetic without the range limit (Net: the fp32 engine)
        self.range_fallbacks = 0  # forwards recomputed because a layer left the 16-bit planes' range
        # the fused PointNetConv adds its layer-1 correction (|.| <= w1r_bound) to the hoisted product INSIDE the kernel: weights that
        # could push a product that passed the watch over fp16's maximum are a violation by themselves
        self._static_range = [(f"sa{l}.layer1", f"geometry weights up to {weights.sa[l - 1]['w1r_bound']:.3g}") for l in (1, 2, 3)
                              if self.prec == PREC_F16X3 and self.options.range_guard and weights.sa[l - 1].get("w1r_bound", 0.0) > self.W1R_LIMIT]
        self.events = None  # set to a list to record (name, start, end) events per launch
        self.events_grouped = False   # True: (name, start, end, launches) per run of consecutive same-name launches
        self._open = None
        self.stem_out = None
        self._ws = None

    # -- small helpers ------------------------------------------------------------------------
    def _chains(self):
        """Chunk chains in flight inside one feature phase: EngineOptions.res_streams in the stream pipeline (which keeps two whole
        phases in flight), EngineOptions.single_res_streams in a lone forward (nothing else fills its round tails)."""
        return max(1, int(self.single_res_streams if getattr(self, "_lone", False) else self.res_streams))

    def _side_stream(self, cur):
        """The second chunk-chain stream of the feature phase running on `cur` (one per feature stream: phases in flight on
        different streams must not meet on a shared side stream)."""
        pool = self.__dict__.setdefault("_side_streams", {})
        key = cur.cuda_stream
        if key not in pool:
            pool[key] = torch.cuda.Stream(priority=int(self.res_priority))
        return pool[key]

    def _call(self, name, fn, *args):
        ev = self.events
        if ev is None:
            check(fn(*args, _lib.stream()), name)
            return
        if self.events_grouped:
            # one HIP-event pair per RUN of consecutive launches of the same name (e.g. the four GEMMs of a residual chunk):
            # an event record between two kernels costs ~8 us of dispatch latency that a per-launch bracket counts as
            # kernel time; entries are [name, start, end, launches], the open run is closed by the next name / flush_events
            if self._open is not None and self._open[0] == name:
                check(fn(*args, _lib.stream()), name)
                self._open[3] += 1
                return
            self.flush_events()
            s = torch.cuda.Event(enable_timing=True)
            s.record()
            check(fn(*args, _lib.stream()), name)
            self._open = [name, s, None, 1]
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        check(fn(*args, _lib.stream()), name)
        e.record()
        ev.append((name, s, e))

    def flush_events(self):
        if self._open is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._open[2] = e
            self.events.append(tuple(self._open))
            self._open = None

    def _gemm(self, name, A, lda, M, lin: Linear, out, ldo, residual=None, ldr=0):
        ep = Epilogue(ptr(lin.bias), ptr(lin.sc0), ptr(lin.sh0), ptr(lin.sc1), ptr(lin.sh1), ptr(residual), ldr,
                      lin.relu0, lin.relu1, lin.relu2, lin.relu_final)
        self._call(name, lib().p2w_gemm, ptr(A), lda, ptr(lin.w), M, lin.N, lin.K, C.byref(ep), ptr(out), ldo)

    TABLE_CELLS_MAX = 1 << 26          # 64 M entries x 20 B = 1.3 GB of workspace at most
    TABLE_SCALE_MAX = 64               # growth factor cap of a level's table after overflows
    TABLE_REST = 64                    # a level that overflows even at that scale (voxels > 9 m, non-finite coordinates) takes the
                                       # sort for this many batches before the table is tried again: a workload of such batches
                                       # pays the discarded geometry pass once per 64 batches instead of on every one

    def _table_cells(self, level, B, N):
        """Table capacity (entries) for the sampler of `level`, 0 = use the sort.  Provision: B voxels x the cells of a 2.3 m cube
        at that resolution (+ margin), times the level's growth factor (x 8 after every overflow, i.e. twice the extent per
        axis; a batch for which that does not fit TABLE_CELLS_MAX takes the sort).  The kernels clear, scan and compact only the part of the table the batch's real
        grid uses, so a generous capacity costs (almost) nothing - what costs is the GRID: it grows with the number of voxels,
        the sort with the number of points.  Many small voxels (B in the hundreds, a few hundred points each) mean tens of
        millions of cells for half a million points, so the table is only taken while the grid of B nominal voxels has at
        most ``table_cells_per_point`` cells per point.  Decided per batch; only the growth factor is remembered."""
        if self.sampler != "table":
            return 0
        if self._table_rest[level] > 0:
            self._table_rest[level] -= 1
            return 0
        per_voxel = (int(2.3 / SA_RES[level]) + 3) ** 3
        if B * per_voxel > self.table_cells_per_point * max(N, 1):
            return 0
        # the growth factor this level has needed so far (x 8 per overflow).  If a table of that extent does not fit for THIS batch's
        # voxel count, take the sort: a smaller table is the one that already overflowed (round 4: a 67-voxel batch of 4 m plot
        # voxels shrank the factor back to 1, overflowed, and paid a discarded geometry pass + the sort on every forward)
        cells = B * per_voxel * self._table_scale[level]
        if cells <= self.TABLE_CELLS_MAX:
            return cells
        # The remembered extent does not fit for THIS many voxels.  The factor is one number per level, but batches differ: after a
        # batch of 4 m plot voxels (factor 8) every later batch of many NOMINAL voxels would take the sort for the life of the engine
        # (ADVICE r4).  So such a batch PROBES the largest smaller factor that fits - a probe that overflows costs one discarded
        # geometry pass, is not answered by growing the factor, and silences probing for TABLE_REST batches (they take the sort).
        if self._table_probe_rest[level] > 0:
            self._table_probe_rest[level] -= 1
            return 0
        s = self._table_scale[level]
        while s > 1 and B * per_voxel * s > self.TABLE_CELLS_MAX:
            s //= 8
        cells = B * per_voxel * s
        if cells > self.TABLE_CELLS_MAX:
            return 0
        self._table_probe[level] = True
        return cells

    def _table_workspace(self, n, cells, device):
        need = int(lib().p2w_voxel_sample_table_ws_bytes(n, cells))
        if need == 0:
            raise RuntimeError("p2w_voxel_sample_table_ws_bytes failed")
        if self._ws_t is None or self._ws_t.numel() < need or self._ws_t.device != device:
            self._ws_t = torch.empty(need, dtype=torch.uint8, device=device)
            # the sampler's between-calls state (p2w_voxel_sample_table_prepared): once per workspace, on the stream that will use it
            check(lib().p2w_voxel_sample_table_prepare(ptr(self._ws_t), self._ws_t.numel(), _lib.stream()), "voxel_sample_table_prepare")
        return self._ws_t

    def _workspace(self, n, device):
        need = int(lib().p2w_voxel_sample_ws_bytes(n))
        if need == 0:
            raise RuntimeError("p2w_voxel_sample_ws_bytes failed (no usable HIP device?)")
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    # -- phase 1 ------------------------------------------------------------------------------
    def _geometry_async(self, pos, reflectance, ptr0, sf, force_sort: bool = False, search_stream=None, defer_searches: bool = False) -> Geometry:
        """Enqueues the geometry phase: first the SAMPLING chain (record packing, the three grid sub-samplings and level
        positions - everything the level sizes depend on - and the one device-to-host copy of those sizes), then the six
        SEARCHES.  ``search_stream`` None: all on the current stream.  Otherwise (the single-call forward) the searches go to
        that stream behind the sampling chain and every search records an event (``geo.ev_nbr[l]``, ``geo.ev_fp[f]``) that the
        feature phase waits for where it first reads the result: the searches of level l + 1 then run BESIDE the features of
        level l.  All buffers are allocated on the current stream either way.  ``defer_searches``: the caller enqueues them
        itself (``geo.launch_searches()``) - the lone forward puts its size-independent feature kernels in front of them."""
        L = lib()
        dev = pos.device
        N, B, k = pos.shape[0], sf.numel(), self.k
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        geo = Geometry(B=B, N=N, k=k, sf=sf)
        geo.args = (pos, reflectance, ptr0, sf)
        geo.stream = torch.cuda.current_stream()
        geo.search_stream = search_stream
        # the three levels' CSR arrays and the table sampler's status words live in ONE buffer, so that the level sizes
        # (ptr_l[B]) and the status reach the host in one copy without a gather kernel in front of it
        pstride = (B + 1 + 3) // 4 * 4
        meta = torch.empty(3 * pstride + 4, **i32)
        status = meta[3 * pstride:]       # per level: 1 = the table sampler's grid did not fit (results undefined)
        geo.table_levels = []
        geo.table_probes = []      # levels whose table was handed out BELOW the remembered growth factor (see _table_cells)
        xyzr0, batch0 = torch.empty((N, 4), **f32), torch.empty(N, **i32)
        self._call("pack_xyzr", L.p2w_pack_xyzr, ptr(pos), pos.stride(0), ptr(reflectance), ptr(ptr0), B, N,
                   ptr(xyzr0), ptr(batch0))
        geo.levels.append(Level(xyzr=xyzr0, ptr=ptr0, batch=batch0, n=N))
        ws = self._workspace(N, dev)
        nbox = int(L.p2w_tile_bbox_count(B, N))
        bbox = {}   # level -> tile bounding boxes of that level's records (kNN pruning; results unchanged)

        def boxes(level):
            if level not in bbox:
                t = torch.empty((nbox, 6), **f32)
                lv_ = geo.levels[level]
                self._call("tile_bbox", L.p2w_tile_bbox, ptr(lv_.xyzr), ptr(lv_.ptr), B, N, ptr(t))
                bbox[level] = t
            return bbox[level]
        grid_search = self.search != "brute"   # brute: whole-voxel streaming kernels (A/B, tests)
        sorted0 = skeys0 = None   # level 0 in cell order (the level-1 sampler's sort), each record carrying its own index
        ckeys, grids, ranks = {}, {}, {}     # level -> cell key of every record (ascending) / p2w_grid of the sampling call
        cstart, cstart0 = {}, None           # level -> cell -> position table of its records (table sampler only)
        aux0 = []
        # ---- sampling chain: sampler l -> level positions l -> sampler l + 1 ... (model.py:103-106,122-126)
        for l, res in enumerate(SA_RES):
            src = geo.levels[l]
            lv = Level(xyzr=torch.empty((N, 4), **f32), ptr=meta[l * pstride: l * pstride + B + 1], batch=torch.empty(N, **i32),
                       idx=torch.empty(N, **i32), nbr=torch.empty((N, k), **i32), deg=torch.empty(N, **i32))
            order = torch.empty(N, **i32) if l == 0 else None
            skeys = torch.empty(N, **i64) if l == 0 else None
            ckeys[l + 1], grids[l + 1] = torch.empty(N, **i64), torch.empty(8, **i64)
            # rank of every source point's cell among the sampled level (= index of its representative): seeds the
            # interpolation searches (level 0 takes part in its sorted order, so it needs the rank per sorted position)
            ranks[l] = torch.empty(N, **i32) if (grid_search and self.fp_hints) else None
            self._table_probe[l] = False
            cells = 0 if force_sort else self._table_cells(l, B, N)
            if self._table_probe[l]:
                geo.table_probes.append(l)
            if cells:
                ws_t = self._table_workspace(N, cells, dev)
                geo.table_levels.append(l)
                # cell -> position tables of the level this call produces (and, for level 0, of its cell-sorted input): the grid
                # searches INTO these candidates look their runs up instead of bisecting the keys
                cstart[l + 1] = torch.empty(cells + 1, **i32) if self.search_index else None
                cstart0 = torch.empty(cells + 1, **i32) if (self.search_index and l == 0) else None
                self._call("voxel_sample", L.p2w_voxel_sample_table_prepared if self.table_prepared else L.p2w_voxel_sample_table, ptr(src.xyzr), ptr(src.ptr), B, N, res, ptr(lv.idx),
                           ptr(lv.ptr), ptr(lv.batch), ptr(order), ptr(skeys), ptr(ckeys[l + 1]), ptr(grids[l + 1]),
                           ptr(ranks[l]) if l > 0 else None, ptr(ranks[l]) if l == 0 else None, ptr(cstart[l + 1]),
                           ptr(cstart0), ptr(status[l:]), cells, ptr(ws_t), ws_t.numel())
            else:
                self._call("voxel_sample", L.p2w_voxel_sample, ptr(src.xyzr), ptr(src.ptr), B, N, res, ptr(lv.idx),
                           ptr(lv.ptr), ptr(lv.batch), ptr(order), ptr(skeys), ptr(ckeys[l + 1]), ptr(grids[l + 1]),
                           ptr(ranks[l]) if l > 0 else None, ptr(ranks[l]) if l == 0 else None, ptr(ws), ws.numel())
            if l == 0:
                # The input points arrive in arbitrary order; the searches that touch level 0 (the ball query as
                # candidates, the last interpolation as queries) run over the cell-sorted copy so that a workgroup's
                # queries are neighbours in space and only the grid rows near them are visited.  Results are unchanged.
                sorted0, skeys0 = torch.empty((N, 4), **f32), skeys
                self._call("index_records", L.p2w_index_records, ptr(src.xyzr), ptr(order), ptr(src.ptr), B, N, ptr(sorted0))
                geo.order, geo.sorted0 = order, sorted0
                geo.rows0_sorted = bool(grid_search and self.fp1_cell_order and self.prec is not None)
                if geo.rows0_sorted:   # position of every input point in the cell order (the P row of a level-0 source point)
                    geo.order64 = order.long()
                    geo.inv0 = torch.empty(N, **i32)
                    geo.inv0[geo.order64] = torch.arange(N, **i32)
                aux0 += [order, sorted0, skeys0]
            self._call("level_gather", L.p2w_level_gather, ptr(src.xyzr), ptr(lv.idx), ptr(lv.batch), ptr(lv.ptr), B, N,
                       ptr(sf), ptr(lv.xyzr))
            geo.levels.append(lv)
        # the level sizes (and the table sampler's status words) start their way to the host here, ahead of the searches
        geo.counts_dev = meta
        geo.counts_host = torch.empty(meta.numel(), dtype=torch.int32, pin_memory=True)
        geo.counts_host.copy_(meta, non_blocking=True)
        geo.pstride = pstride
        geo.sizes_ready = torch.cuda.Event()
        geo.sizes_ready.record()
        # ---- searches (as a closure: the lone forward enqueues its size-independent feature kernels between the two chains)
        geo.ev_nbr, geo.ev_fp = {}, {}

        def searching():
            return torch.cuda.stream(search_stream) if search_stream is not None else contextlib.nullcontext()

        def mark(table, key):
            if search_stream is not None:
                table[key] = torch.cuda.Event()
                table[key].record()

        def sa_search(l):
            res = SA_RES[l]
            src, lv = geo.levels[l], geo.levels[l + 1]
            if l == 0:   # model.py:117-118: the 0.04 level uses radius(r = 2*resolution)
                if grid_search:
                    self._call("ball_query", L.p2w_ball_query_grid_indexed, ptr(sorted0), ptr(skeys0), ptr(src.ptr), ptr(grids[1]),
                               ptr(cstart0), ptr(src.xyzr), ptr(lv.idx), ptr(lv.ptr), B, N, res * 2, k, ptr(lv.nbr), ptr(lv.deg),
                               SEARCH_X_INDEX_IN_W | (SEARCH_BOX if self.search_box & 1 else 0))
                else:
                    box0 = torch.empty((nbox, 6), **f32)
                    self._call("tile_bbox", L.p2w_tile_bbox, ptr(sorted0), ptr(src.ptr), B, N, ptr(box0))
                    self._call("ball_query", L.p2w_ball_query, ptr(sorted0), ptr(src.ptr), ptr(src.xyzr), ptr(lv.idx),
                               ptr(lv.ptr), B, N, res * 2, k, ptr(lv.nbr), ptr(lv.deg), ptr(box0), SEARCH_X_INDEX_IN_W)
                    geo.aux.append(box0)
            elif grid_search:   # model.py:120
                self._call("knn", L.p2w_knn_grid_indexed, ptr(src.xyzr), ptr(ckeys[l]), ptr(src.ptr), ptr(grids[l]), ptr(cstart.get(l)),
                           ptr(src.xyzr),
                           ptr(lv.idx), ptr(lv.ptr), B, N, k, ptr(lv.nbr), ptr(lv.deg), None,
                           (SEARCH_BOX if self.search_box & 2 else 0))
            else:
                self._call("knn", L.p2w_knn, ptr(src.xyzr), ptr(src.ptr), ptr(src.xyzr), ptr(lv.idx), ptr(lv.ptr), B, N,
                           k, ptr(lv.nbr), ptr(lv.deg), ptr(boxes(l)), 0)
            mark(geo.ev_nbr, l + 1)

        def fp_search(f):
            # k=2 searches of knn_interpolate (model.py:149): fine level f queries coarse level f+1
            fine, coarse = geo.levels[f], geo.levels[f + 1]
            nbr, deg = torch.empty((N, 2), **i32), torch.empty(N, **i32)
            # level 0 queries run over the cell-sorted copy; their result rows go to the points' own rows (row in .w) - or stay in
            # cell order when the feature phase keeps level 0 in that order (fp1_cell_order)
            q, fl = (sorted0, 0 if getattr(geo, "rows0_sorted", False) else SEARCH_Q_ROW_IN_W) if f == 0 else (fine.xyzr, 0)
            if grid_search:
                hint = None
                if ranks.get(f) is not None:   # both of a point's two nearest coarse points are within its cell
                    hint = torch.empty(N, **f32)   # representative's / a storage neighbour's representative's distance
                    self._call("knn_hint", L.p2w_knn_hint2, ptr(q), ptr(ranks[f]), ptr(fine.ptr), B, N, ptr(coarse.xyzr),
                               ptr(hint))
                    geo.aux.append(hint)
                self._call("knn2", L.p2w_knn_grid_indexed, ptr(coarse.xyzr), ptr(ckeys[f + 1]), ptr(coarse.ptr), ptr(grids[f + 1]),
                           ptr(cstart.get(f + 1)), ptr(q), None, ptr(fine.ptr), B, N, 2, ptr(nbr), ptr(deg), ptr(hint),
                           fl | (SEARCH_BOX if self.search_box & 4 else 0))
            else:
                self._call("knn2", L.p2w_knn, ptr(coarse.xyzr), ptr(coarse.ptr), ptr(q), None, ptr(fine.ptr), B, N, 2,
                           ptr(nbr), ptr(deg), ptr(boxes(f + 1)), fl)
            geo.fp_nbr[f] = (nbr, deg)
            mark(geo.ev_fp, f)
        jobs = [lambda l=l: sa_search(l) for l in range(3)] + [lambda f=f: fp_search(f) for f in (2, 1, 0)]

        def launch_searches(count=None, after=None):
            """Enqueues the next `count` searches (all that are left: None) in the order ball query, kNN level 2, kNN level 3, the
            interpolation searches of levels 2, 1, 0; `after`: an event the search stream waits for first (the staggered lone
            forward holds a search back until the feature kernel it should run beside has started)."""
            first = len(jobs) == 6
            todo = jobs[: (len(jobs) if count is None else count)]
            del jobs[: len(todo)]
            if first and search_stream is not None:
                search_stream.wait_event(geo.sizes_ready)
            if after is not None and search_stream is not None:
                search_stream.wait_event(after)
            with searching():
                for job in todo:
                    job()
                if not jobs:
                    geo.done = torch.cuda.Event()
                    geo.done.record()
            if not jobs:
                geo.launch_searches = None
                geo.aux += list(bbox.values())
        aux0 += list(ckeys.values()) + list(grids.values()) + [t for t in ranks.values() if t is not None]
        aux0 += [t for t in list(cstart.values()) + [cstart0] if t is not None]
        if getattr(geo, "rows0_sorted", False):
            aux0 += [geo.inv0, geo.order64]
        geo.aux = aux0
        geo.launch_searches = launch_searches
        if not defer_searches:
            launch_searches()
        return geo

    def _geometry_finish(self, geo):
        """The host's wait for the three level sizes (in flight since the end of the sampling chain, while the searches run)."""
        geo.sizes_ready.synchronize()
        ps = geo.pstride
        overflow = [l for l in geo.table_levels if int(geo.counts_host[3 * ps + l])]
        if overflow:
            # the batch's cell grid did not fit the sampler's table at these levels (device-side knowledge): everything
            # downstream of the first of them is undefined.  Repeat the geometry with the sort (rare: voxels much larger than
            # 2 m), and give the table 8 x the room next time.
            for l in overflow:   # more room next time (capped); whether a batch's table is worth taking is _table_cells' decision
                if l in getattr(geo, "table_probes", ()):   # a probe below the remembered factor failed: no growth, no probing for a while
                    self._table_probe_rest[l] = self.TABLE_REST
                    continue
                if self._table_scale[l] >= self.TABLE_SCALE_MAX:
                    self._table_rest[l] = self.TABLE_REST     # no more room to give: the sort serves the next batches
                self._table_scale[l] = min(self._table_scale[l] * 8, self.TABLE_SCALE_MAX)
            search_stream = getattr(geo, "search_stream", None)
            with torch.cuda.stream(geo.stream):
                if search_stream is not None:       # the discarded searches still read the buffers this Geometry is about to drop
                    if geo.launch_searches is not None:
                        geo.launch_searches()
                    geo.stream.wait_event(geo.done)
                redo = self._geometry_async(*geo.args, force_sort=True, search_stream=search_stream)
            redo.sizes_ready.synchronize()
            geo.__dict__.update(redo.__dict__)
        for l in (1, 2, 3):
            geo.levels[l].n = int(geo.counts_host[(l - 1) * geo.pstride + geo.B])
        return geo

    def geometry(self, pos, reflectance, ptr0, sf) -> Geometry:
        return self._geometry_finish(self._geometry_async(pos, reflectance, ptr0, sf))

    # -- phase 2 ------------------------------------------------------------------------------
    def features(self, geo: Geometry, keep: dict | None = None):
        """Enqueues the feature phase on the current stream.  With the range guard on, ``geo.watch`` then holds the phase's range
        report (in flight on the same stream: ``checked`` evaluates it once the phase has finished)."""
        if self.prec is not None:
            if getattr(geo, "early", None) is None:      # (an early part has opened the range watch already)
                self._range_begin(geo.sf.device)
            logits = self._features_h2(geo, keep)
            geo.watch = self._range_end()
            return logits
        geo.watch = None
        if getattr(geo, "search_stream", None) is not None:   # the fp32 parity mode runs behind the searches
            if geo.launch_searches is not None:
                geo.launch_searches()
            torch.cuda.current_stream().wait_event(geo.done)
        return self._features_fp32(geo, keep)

    def checked(self, geo: Geometry, logits, keep: dict | None = None):
        """The range guard's verdict on a FINISHED feature phase (the caller has waited for it): `logits`, or - when a layer's
        outputs left the range the 16-bit planes carry - the logits of ``fallback`` (the fp32 MFMA engine) on the same geometry,
        enqueued on the current stream.  Silent saturation is never an outcome: without a fallback the forward raises."""
        bad = self.range_violations(getattr(geo, "watch", None))
        if not bad:
            return logits
        what = "; ".join(f"{n}: {v}" for n, v in bad[:4])
        # "no value above 2^-5" is SAMPLED (wave 0 of every persistent workgroup's first tile): on a small or sparse layer the
        # sample can miss.  It is acted on where a recomputation is available; it never raises by itself.
        sampled_only = all(v.startswith("no value above") for _, v in bad)
        if sampled_only and (self.fallback is None or getattr(geo, "rows0_sorted", False)):
            if not getattr(self, "_seen_warned", False):
                import warnings
                self._seen_warned = True
                warnings.warn(f"pointstowood_amd: precision {self.precision!r}: the range watch saw no value above {self.RANGE_LO:g} in its "
                              f"sample of {what} and no fp32 recomputation is available in this configuration: logits returned as computed")
            return logits
        if self.fallback is None:
            raise RuntimeError(f"precision {self.precision!r}: activations outside the range of the 16-bit planes ({what}); "
                               "use precision='fp32'")
        if self.range_fallbacks == 0:
            import warnings
            warnings.warn(f"pointstowood_amd: precision {self.precision!r} cannot carry this checkpoint's activations ({what}): such "
                          "batches are recomputed on the fp32 MFMA path (about 4x slower)")
        self.range_fallbacks += 1
        return self.fallback(geo, keep)

    # -- range watch (EngineOptions.range_guard) -------------------------------------------------------------
    RANGE_SLOTS = 64                       # layers of a forward
    RANGE_WORDS = 1024                     # P2W_RANGE_WORDS: words of one launch's report (16 slots of (over, seen), 64 words apart)
    RANGE_HI, RANGE_LO = 6.0e4, 2.0 ** -5  # the thresholds behind the two bits (P2W_RANGE_HI / _LO)
    W1R_LIMIT = 65504.0 - 6.0e4            # room the PointNetConv's layer-1 correction has above a hoisted product that passed the watch

    def _watch(self, layer, rows=1):
        """Device address of the range-watch word of `layer` (one per layer of the forward in flight), or None (guard off, or a
        launch without rows: it reports nothing)."""
        w = self._range
        if w is None or layer is None or rows <= 0:
            return None
        names, buf = w
        if layer not in names:
            if len(names) >= self.RANGE_SLOTS:
                return None
            names.append(layer)
        return buf.data_ptr() + 4 * self.RANGE_WORDS * names.index(layer)   # one report block per layer

    def _range_begin(self, dev):
        self._range = ([], torch.zeros(self.RANGE_SLOTS * self.RANGE_WORDS, dtype=torch.int32, device=dev)) if (self.range_guard and self.prec == PREC_F16X3) else None

    def _range_end(self):
        """(layer names, device words, pinned host copy in flight on the current stream) of the forward just enqueued."""
        w, self._range = self._range, None
        if w is None:
            return None
        host = torch.empty(2 * self.RANGE_SLOTS, dtype=torch.int32, pin_memory=True)
        # OR over a layer's report slots on the device (one small reduction), then (over, seen) per layer to the host
        host.copy_(w[1].view(self.RANGE_SLOTS, self.RANGE_WORDS // 64, 64)[:, :, :2].amax(dim=1).reshape(-1), non_blocking=True)
        return w[0], w[1], host

    def range_violations(self, watch):
        """Layers of a FINISHED forward whose outputs left the range the 16-bit planes carry: [(layer, what)]."""
        if watch is None:
            return []
        names, _, host = watch
        bad = list(self._static_range)
        flags = host[: 2 * len(names)].tolist()
        for i, n in enumerate(names):
            if flags[2 * i]:
                bad.append((n, f"values beyond +-{self.RANGE_HI:g} (or NaN)"))
            elif not flags[2 * i + 1]:
                bad.append((n, f"no value above {self.RANGE_LO:g}"))
        return bad

    def _gemm_h2(self, name, A, ldh_a, M, lin: Linear, out_f32=None, ldo=0, out_h2=None, ldh_o=0, residual=None, ldr=0,
                 residual_h=False, watch=None, interp=None):
        ep = Epilogue(ptr(lin.bias), ptr(lin.sc0), ptr(lin.sh0), ptr(lin.sc1), ptr(lin.sh1), ptr(residual), ldr,
                      lin.relu0, lin.relu1, lin.relu2, lin.relu_final, self._watch(watch, M),
                      ptr(interp), 0 if interp is None else residual.shape[0])
        flags = self.gemm_flags | (GEMM_RESIDUAL_H if residual_h else 0)
        if self.gemm_stream_k:
            ws = self._sk_workspace(A.device)
            self._call(name, lib().p2w_gemm_h2_sk, self.prec, ptr(A), ldh_a, ptr(lin.w16), lin.wscale, M, lin.N, lin.K, C.byref(ep),
                       ptr(out_f32), ldo, ptr(out_h2), ldh_o, ptr(ws), ws.numel(), flags)
            return
        self._call(name, lib().p2w_gemm_h2, self.prec, ptr(A), ldh_a, ptr(lin.w16), lin.wscale, M, lin.N, lin.K, C.byref(ep),
                   ptr(out_f32), ldo, ptr(out_h2), ldh_o, flags)

    def _sk_workspace(self, dev):
        """The stream-K workspace of the CURRENT stream (launches on different streams may overlap, so each stream that runs
        GEMMs owns one; allocated once, p2w_gemm_h2_sk_ws_bytes() = 32 MiB on a 256-CU chip)."""
        pool = self.__dict__.setdefault("_sk_ws", {})
        key = _lib.stream()
        ws = pool.get(key)
        if ws is None:
            ws = pool[key] = torch.empty(int(lib().p2w_gemm_h2_sk_ws_bytes()), dtype=torch.uint8, device=dev)
        return ws

    def _hoist(self, xh_src, pitch_src, n_src, p, l):
        """SA level l's hoisted layer-1 product P = x_src W1x^T + b1: n_src rows + one all-zero row (read by empty neighbour slots),
        row pitch padded to whole K slabs with zero columns (p2w_sa_conv_h reads it with unconditional loads)."""
        ka = 32 if self.prec == PREC_F16X3 else 64
        C1 = p["C1"]
        C1p = (C1 + ka - 1) // ka * ka
        P = torch.empty((n_src + 1, C1p), dtype=torch.float32, device=xh_src.device)     # (row n_src: zeroed by p2w_sa_conv_h's pre-pass)
        if C1p != C1:
            P[:, C1:].zero_()
        self._gemm_h2("gemm_hoist", xh_src, pitch_src, n_src, p["hoist"], out_f32=P, ldo=C1p, watch=f"hoist{l}")
        return P

    def _features_early(self, geo: Geometry):
        """The part of the H feature phase that needs the input points only (stem, SA1's hoisted layer-1 product): the
        single-call forward enqueues it BEFORE the host waits for the level sizes, so the GPU is never idle during that wait."""
        L, w = lib(), self.w
        dev = geo.sf.device
        Cw, N, prec = w.C, geo.N, self.prec
        ka, planes = (32, 2) if prec == PREC_F16X3 else (64, 1)
        hdt = torch.bfloat16 if self.precision == "bf16" else torch.float16
        F3 = 16 * Cw
        pitch0 = (F3 + Cw + ka - 1) // ka * ka
        cat0 = torch.empty((N, planes * pitch0), dtype=hdt, device=dev)      # rows of FP1: [interpolated (16 C) | stem features (C)]
        xh0 = cat0[:, planes * F3:]
        x0 = torch.empty((N, Cw), dtype=torch.float32, device=dev)
        # Level 0 in the sampler's CELL ORDER (fp1_cell_order): the H rows of the input points' features - the skip columns of FP1's rows,
        # the A operand of SA1's hoisted product - are row p = the p-th point of the cell-sorted order, so are FP1's rows, its MLP, the head;
        # the logits are scattered back at the end.  Spatial neighbours are then memory neighbours: the last interpolation's two coarse
        # rows per point and SA1's P rows come from L2.  The fp32 stem features (data.x, model.py:228) stay in input order.
        if bool(getattr(geo, "rows0_sorted", False)):
            self._call("stem", L.p2w_stem_h2_indexed, prec, ptr(geo.sorted0), N, ptr(w.stem_w), ptr(w.stem_b), Cw, ptr(x0), ptr(xh0),
                       pitch0, self._watch("stem", N))
        else:
            self._call("stem", L.p2w_stem_h2, prec, ptr(geo.levels[0].xyzr), N, ptr(w.stem_w), ptr(w.stem_b), Cw, ptr(x0), ptr(xh0),
                       pitch0, self._watch("stem", N))
        self.stem_out = x0
        P1 = self._hoist(xh0, pitch0, N, w.sa[0], 1)
        return dict(lv0=geo.levels[0], cat0=cat0, x0=x0, P1=P1)

    def _features_h2(self, geo: Geometry, keep: dict | None = None):
        """H pipeline (f16x3 / fp16 / bf16): every GEMM operand is an H tensor (16-bit planes: fp16 hi/lo for f16x3, one
        fp16 / bf16 plane otherwise) written once by its producer.

        Skip connections are concatenated IN PLACE: the rows the first layer of FP module f reads are [interpolated coarse
        features (Fc) | skip features (Fs)] (model.py:149-151); the buffer of every such level exists from the start and the
        producer of the skip features (the stem, the residual blocks' last layer) writes its H output straight into the skip
        columns, so nothing is copied later and no fp32 twin of those features is written.  In f16x3 the residual blocks also
        read their residual from the H tensor their first layer consumes (hi + lo = the fp32 value to 2^-22) instead of an
        fp32 copy.  ``keep`` (tests, profiling) additionally asks for the fp32 forms."""
        L, w = lib(), self.w
        dev = geo.sf.device
        Cw = w.C
        new = lambda r, c: torch.empty((r, c), dtype=torch.float32, device=dev)
        prec = self.prec
        ka, planes = (32, 2) if prec == PREC_F16X3 else (64, 1)
        hdt = torch.bfloat16 if self.precision == "bf16" else torch.float16
        pad8 = lambda f: (f + ka - 1) // ka * ka   # H row pitch: zero-padded to the GEMM's K slab so it can be DMA-staged
        newh = lambda r, f: torch.empty((r, planes * pad8(f)), dtype=hdt, device=dev)
        hcol = lambda t, c: t[:, planes * c:]      # view of an H tensor from column c on (c a multiple of the slab width)
        lv, N, B = geo.levels, geo.N, geo.B
        F3 = 16 * Cw
        res_h = prec == PREC_F16X3                 # residual read from H (the single-plane modes keep their fp32 residual)
        # concatenated rows of the four FP modules: fine level f = 0..3 gets [m_f, Fc + Fs_f], Fc = 16 C interpolated columns
        Fs = [Cw, 4 * Cw, 8 * Cw, 16 * Cw]
        rows = [N, lv[1].n, lv[2].n, lv[3].n]
        pitch = [pad8(F3 + Fs[f]) for f in range(4)]
        early = getattr(geo, "early", None) or self._features_early(geo)
        cat = [early["cat0"]] + [newh(rows[f], F3 + Fs[f]) for f in range(1, 4)]
        xh = [hcol(cat[f], F3) for f in range(4)]   # H features of level f = the skip columns of its FP module's rows
        x0 = early["x0"]
        sorted0 = bool(getattr(geo, "rows0_sorted", False))
        wait = lambda ev: torch.cuda.current_stream().wait_event(ev) if ev is not None else None   # a search result is about to be read
        ev_nbr, ev_fp = getattr(geo, "ev_nbr", {}), getattr(geo, "ev_fp", {})

        def need(table, key):
            """The search result (table, key) is about to be read: if the staggered forward has not enqueued that search yet, it
            goes out now; then the caller's stream waits for its event."""
            while table.get(key) is None and getattr(geo, "launch_searches", None) is not None:
                geo.launch_searches(1)
            wait(table.get(key))

        def stage(count):
            """Staggered lone forward: enqueue the next `count` searches behind the feature kernel just launched."""
            if getattr(geo, "launch_searches", None) is not None and getattr(geo, "search_stream", None) is not None:
                ev = torch.cuda.Event()
                ev.record()
                geo.launch_searches(count, after=ev)
        if keep is not None:
            keep["stem"] = x0
        x3 = None
        for l in (1, 2, 3):
            p, src, dst = w.sa[l - 1], lv[l - 1], lv[l]
            M, C1, C2, E = dst.n, p["C1"], p["C2"], 4 * p["C2"]
            # hoisted layer-1 product P = x_src W1x^T + b1: src.n rows + one all-zero row (read by empty neighbour slots), row
            # pitch padded to whole K slabs with zero columns (p2w_sa_conv_h reads it with unconditional loads)
            C1p = pad8(C1)
            P = early["P1"] if l == 1 else self._hoist(xh[l - 1], pitch[l - 1], src.n, p, l)
            need(ev_nbr, l)
            conv = new(M, C2) if (keep is not None or not res_h) else None
            convh = newh(M, C2)
            # level 1 is the ball query: on sparse input most targets have few neighbours, and those with <= 8 share an MFMA
            # tile four at a time (P2W_SA_PACK8); the kNN levels always fill their 32 slots
            sa_flags = self.sa_flags | (SA_PACK8 if (l == 1 and self.sa_pack) else 0)
            meta = torch.empty(int(L.p2w_sa_conv_h_ws_bytes(M, sa_flags)) + 65536, dtype=torch.uint8, device=dev)   # per-edge (j, normalised offset) scratch (+ room for diagnostics)
            if keep is not None:
                keep[f"sa{l}_module.ws"] = meta
            self._call("sa_conv", L.p2w_sa_conv_h_rows, prec, ptr(P), C1p, src.n, ptr(src.xyzr), ptr(dst.idx), ptr(dst.batch),
                       ptr(geo.sf), ptr(dst.nbr), ptr(dst.deg), geo.k, M, ptr(p["w1r4"]), ptr(p["W2"].w16),
                       p["W2"].wscale, C1, C2, ptr(p["b2"]), ptr(p["bn_s"]), ptr(p["bn_t"]), ptr(conv), C2, ptr(convh),
                       pad8(C2), ptr(meta), meta.numel(), sa_flags, ptr(geo.inv0) if (l == 1 and sorted0) else None, self._watch(f"sa{l}", M))
            stage(1 if l < 3 else 2)     # (no-op unless staggered: kNN level 3 behind SA1, interpolation search 2 behind SA2, 1 and 0 behind SA3)
            # fp32 form of the level's output: level 3 feeds cat(x, pos) of the global module; otherwise only on request
            out = new(M, C2) if (keep is not None or l == 3) else None
            # residual block in row chunks: the two 4F-wide intermediates of a chunk (2 x chunk x 4F x 4 B) are written
            # and re-read while they still sit in the 256 MiB Infinity Cache instead of round-tripping HBM
            chunk = self.res_chunk_rows if self.res_chunk_rows > 0 else M
            chunk = max(256, min(M, pick_chunk(M, chunk * 512 // E, E // 256, full_rounds=self.chunk_full_rounds and not self.gemm_stream_k) if self.chunk_pick
                                 else (chunk * 512 // E) // 256 * 256))   # ~same bytes per chunk at every level
            # Chunks are independent chains of four GEMMs; alternating them between two streams lets the tiles of one
            # chain fill the CUs the other leaves idle at its wave tails (a 1122-row tail chunk at level 3 otherwise
            # runs four GEMMs at 16 % chip fill).
            nst = max(1, min(self._chains(), -(-M // chunk)))
            cur = torch.cuda.current_stream()
            side = self._side_stream(cur) if nst > 1 else None
            lanes = [cur] + ([side] if nst > 1 else [])
            bufs = [(newh(min(M, chunk), E), newh(min(M, chunk), E)) for _ in lanes]
            if nst > 1:
                ready = torch.cuda.Event()
                ready.record(cur)
                side.wait_event(ready)
            for ci, r0 in enumerate(range(0, M, chunk)):
                m = min(chunk, M - r0)
                (e1, e2), st = bufs[ci % len(lanes)], lanes[ci % len(lanes)]
                with torch.cuda.stream(st):
                    self._gemm_h2("gemm_res", convh[r0:], pad8(C2), m, p["g1"], out_h2=e1, ldh_o=pad8(E), watch=f"res{l}.1")
                    self._gemm_h2("gemm_res", e1, pad8(E), m, p["g2"], out_h2=e2, ldh_o=pad8(E), watch=f"res{l}.2")
                    self._gemm_h2("gemm_res", e2, pad8(E), m, p["g3"], out_h2=e1, ldh_o=pad8(E), watch=f"res{l}.3")
                    self._gemm_h2("gemm_res", e1, pad8(E), m, p["g4"], out_f32=None if out is None else out[r0:], ldo=C2,
                                  out_h2=xh[l][r0:], ldh_o=pitch[l],
                                  residual=convh[r0:] if res_h else conv[r0:], ldr=pad8(C2) if res_h else C2, residual_h=res_h,
                                  watch=f"res{l}.4")
            if nst > 1:   # join: everything after this level (and the buffers' reuse) is ordered behind both chains
                done = torch.cuda.Event()
                done.record(side)
                cur.wait_event(done)
            if l == 3:
                x3 = out
            if keep is not None:
                keep[f"sa{l}_module.conv"], keep[f"sa{l}_module.out"] = conv, out
        # GlobalSAModule (model.py:134-140)
        M3 = lv[3].n
        cat_g = newh(M3, F3 + 4)
        self._call("concat_xyz", L.p2w_concat_xyz_h2, prec, ptr(x3), F3, ptr(lv[3].xyzr), M3, ptr(cat_g), pad8(F3 + 4))
        h1, h2 = newh(M3, F3), new(M3, F3)
        self._gemm_h2("gemm_mlp", cat_g, pad8(F3 + 4), M3, w.sa4[0], out_h2=h1, ldh_o=pad8(F3), watch="sa4.0")
        self._gemm_h2("gemm_mlp", h1, pad8(F3), M3, w.sa4[1], out_f32=h2, ldo=F3, watch="sa4.1")
        g = new(B, F3)
        self._call("segment_max", L.p2w_segment_max, ptr(h2), F3, F3, ptr(lv[3].ptr), B, ptr(g))
        if keep is not None:
            keep["sa4_module.out"] = g
        # FPModule 4..1 (model.py:148-153)
        nbr4 = torch.empty(M3, dtype=torch.int32, device=dev)
        deg4 = torch.empty(M3, dtype=torch.int32, device=dev)
        self._call("fill_batch_nbr", L.p2w_fill_batch_nbr, ptr(lv[3].batch), M3, ptr(nbr4), ptr(deg4))
        zc = self.__dict__.setdefault("_zeros_c", {})      # the pooled level's positions (model.py:138: zeros): one constant per batch size
        if (B, dev) not in zc:
            zc[(B, dev)] = torch.zeros((B, 4), dtype=torch.float32, device=dev)
            torch.cuda.current_stream().synchronize()      # (once per batch size: later phases on other streams read it without an event)
        zeros_c = zc[(B, dev)]
        # Row-chunked chains: interpolate -> MLP layer 0 -> MLP layer 1 (-> head for fp1) run chunk by chunk so the wide
        # intermediates of a chunk are consumed out of the Infinity Cache (same trick as the residual blocks).
        y, y_xyzr, y_h = g, zeros_c, None      # coarse features (fp32), their positions, their H form (where the next module hoists)
        y_rows, y_cols = g.shape               # (a module whose successor takes the linearity route leaves no fp32 form)
        logits = torch.empty(N, dtype=torch.float32, device=dev) if w.num_classes == 1 else None
        o_multi = new(N, w.num_classes) if w.num_classes != 1 else None
        logits_out = logits
        for fl in (4, 3, 2, 1):
            fine = lv[fl - 1]
            m, Fc, cf, ld = fine.n, y_cols, cat[fl - 1], pitch[fl - 1]
            nbr, deg, kw = (nbr4, deg4, 1) if fl == 4 else (*geo.fp_nbr[fl - 1], 2)
            if fl < 4:
                need(ev_fp, fl - 1)
            fine_xyzr = geo.sorted0 if (fl == 1 and sorted0) else fine.xyzr   # FP1's rows (and fp_nbr[0]'s) are in cell order then
            l0, l1 = w.fp[fl]
            # layer 0's interpolated half on the coarse rows (PackedWeights.fp_split)?  This module: the previous one left the H
            # form of its output; the next one: this module's output must leave one
            hoists = lambda rows_c, rows_f: bool(self.fp_hoist and 0 < rows_c <= self.fp_hoist_ratio * rows_f)
            Mx = y_rows
            hoist = hoists(Mx, m)
            next_hoist = fl >= 2 and hoists(m, lv[fl - 2].n)
            if hoist:
                li, ls = w.fp_split[fl]
                if y_h is None:      # (module 4: the pooled rows have no H form yet)
                    y_h = newh(Mx, Fc + 4)
                    self._call("concat_xyz", L.p2w_concat_xyz_h2, prec, ptr(y), Fc, ptr(y_xyzr), Mx, ptr(y_h), pad8(Fc + 4))
                    y_ld = pad8(Fc + 4)
                else:
                    y_ld = pad8(Fc)
                Z = new(Mx, l0.N)
                self._gemm_h2("gemm_mlp", y_h, y_ld, Mx, li, out_f32=Z, ldo=l0.N)
                rec = torch.empty((m, 4), dtype=torch.int32, device=dev)
                self._call("interp_concat", L.p2w_interp_weights, ptr(y_xyzr), ptr(fine_xyzr), ptr(nbr), ptr(deg), kw, m, ptr(rec))
            yh_full = newh(m, l1.N) if next_hoist else None
            chunk = self.res_chunk_rows if self.res_chunk_rows > 0 else m
            chunk = max(256, min(m, pick_chunk(m, chunk * 512 // (Fc + Fs[fl - 1]), max(1, l0.N // 256)) if self.chunk_pick
                                 else (chunk * 512 // (Fc + Fs[fl - 1])) // 256 * 256))
            mc = min(m, chunk)
            need_f32 = (fl > 1 and not next_hoist) or keep is not None   # (the next module interpolates the fp32 rows unless it hoists)
            b = new(m, l1.N) if need_f32 else None
            # chunk chains alternate between two streams like the residual blocks' (each lane has its own intermediates)
            nst = max(1, min(self._chains(), -(-m // chunk)))
            cur = torch.cuda.current_stream()
            side = self._side_stream(cur) if nst > 1 else None
            lanes = [cur] + ([side] if nst > 1 else [])
            # head for one class (model.py:241-243): conv1 + BN + ReLU + conv2 as ONE operator, its [m, 512] intermediate never
            # reaches HBM (p2w_gemm_h2_rowdot: per-slice partial dot products in the GEMM's epilogue + a finishing pass)
            one = fl == 1 and w.num_classes == 1
            bufs = [dict(a=newh(mc, l0.N), yh=newh(mc, l1.N) if fl == 1 else None,
                         hws=torch.empty(int(L.p2w_gemm_h2_rowdot_ws_bytes(mc, w.head1.N)), dtype=torch.uint8, device=dev) if one else None,
                         hdh=newh(mc, F3) if (fl == 1 and not one) else None) for _ in lanes]
            if nst > 1:
                ready = torch.cuda.Event()
                ready.record(cur)
                side.wait_event(ready)
            for ci, r0 in enumerate(range(0, m, chunk)):
                mm = min(chunk, m - r0)
                bf, st = bufs[ci % len(lanes)], lanes[ci % len(lanes)]
                a, yh, hws, hdh = bf["a"], bf["yh"], bf["hws"], bf["hdh"]
                with torch.cuda.stream(st):
                    if hoist:   # relu(W_s skip + b + interp(W_i y)): the interpolation of Z's rows happens in the epilogue
                        self._gemm_h2("gemm_mlp", xh[fl - 1][r0:], ld, mm, ls, out_h2=a, ldh_o=pad8(l0.N), residual=Z, ldr=l0.N,
                                      interp=rec[r0:], watch=f"fp{fl}.0")
                    else:
                        # the interpolated part only (skip = NULL): the skip columns of these rows were written by their producer
                        self._call("interp_concat", L.p2w_interp_concat_h2, prec, ptr(y), Fc, ptr(y_xyzr), ptr(fine_xyzr[r0:]),
                                   ptr(nbr[r0:]), ptr(deg[r0:]), kw, None, 0, mm, ptr(cf[r0:]), ld)
                        self._gemm_h2("gemm_mlp", cf[r0:], ld, mm, l0, out_h2=a, ldh_o=pad8(l0.N), watch=f"fp{fl}.0")
                    if yh_full is not None:
                        yh = yh_full[r0:]
                    self._gemm_h2("gemm_mlp", a, pad8(l0.N), mm, l1, out_f32=None if b is None else b[r0:], ldo=l1.N,
                                  out_h2=yh, ldh_o=pad8(l1.N), watch=f"fp{fl}.1")
                    if fl == 1:   # head (model.py:241-243) on the same chunk
                        if one:
                            lin = w.head1
                            ep = Epilogue(ptr(lin.bias), ptr(lin.sc0), ptr(lin.sh0), ptr(lin.sc1), ptr(lin.sh1), None, 0,
                                          lin.relu0, lin.relu1, lin.relu2, lin.relu_final)
                            self._call("gemm_mlp", L.p2w_gemm_h2_rowdot, prec, ptr(yh), pad8(F3), ptr(lin.w16), lin.wscale, mm, lin.N,
                                       lin.K, C.byref(ep), ptr(w.head2_w), float(w.head2_b[0]), ptr(logits[r0:]), ptr(hws),
                                       hws.numel(), self.gemm_flags)
                        else:   # multi-class head: conv2 is one more (narrow) GEMM over the H form of conv1's output
                            self._gemm_h2("gemm_mlp", yh, pad8(F3), mm, w.head1, out_h2=hdh, ldh_o=pad8(F3), watch="head1")
                            self._gemm_h2("gemm_mlp", hdh, pad8(F3), mm, w.head2, out_f32=o_multi[r0:], ldo=w.num_classes)
            if nst > 1:   # join
                done = torch.cuda.Event()
                done.record(side)
                cur.wait_event(done)
            y, y_xyzr, y_h = b, fine.xyzr, yh_full
            y_rows, y_cols = m, l1.N
            if keep is not None:
                if fl == 1 and sorted0 and b is not None:   # rows back in input order for whoever asked
                    bo = torch.empty_like(b)
                    bo[geo.order64] = b
                    b = bo
                keep[f"fp{fl}_module.out"] = b
        if sorted0:   # cell order -> input order (one row per point: 4 B, or num_classes x 4 B)
            if w.num_classes == 1:
                logits_out = torch.empty_like(logits)
                logits_out[geo.order64] = logits
                logits = logits_out
            else:
                om = torch.empty_like(o_multi)
                om[geo.order64] = o_multi
                o_multi = om
        if w.num_classes != 1:
            logits = o_multi.t()
        return torch.squeeze(logits)

    def _features_fp32(self, geo: Geometry, keep: dict | None = None):
        L, w = lib(), self.w
        dev = geo.sf.device
        Cw = w.C
        new = lambda r, c: torch.empty((r, c), dtype=torch.float32, device=dev)
        lv = geo.levels
        N = geo.N
        x = [new(N, Cw)]
        self._call("stem", L.p2w_stem, ptr(lv[0].xyzr), N, ptr(w.stem_w), ptr(w.stem_b), Cw, ptr(x[0]))
        self.stem_out = x[0]
        if keep is not None:
            keep["stem"] = x[0]
        for l in (1, 2, 3):
            p, src, dst = w.sa[l - 1], lv[l - 1], lv[l]
            M, C1, C2, E = dst.n, p["C1"], p["C2"], 4 * p["C2"]
            P = new(src.n, C1)
            self._gemm("gemm_hoist", x[l - 1], p["F_in"], src.n, p["hoist"], P, C1)
            conv = new(M, C2)
            self._call("sa_conv", L.p2w_sa_conv, ptr(P), C1, ptr(src.xyzr), ptr(dst.idx), ptr(dst.batch), ptr(geo.sf),
                       ptr(dst.nbr), ptr(dst.deg), geo.k, M, ptr(p["w1r4"]), ptr(p["W2"].w), C1, C2, ptr(p["b2"]),
                       ptr(p["bn_s"]), ptr(p["bn_t"]), ptr(conv), C2)
            e1, e2 = new(M, E), new(M, E)
            self._gemm("gemm_res", conv, C2, M, p["g1"], e1, E)
            self._gemm("gemm_res", e1, E, M, p["g2"], e2, E)
            self._gemm("gemm_res", e2, E, M, p["g3"], e1, E)
            out = new(M, C2)
            self._gemm("gemm_res", e1, E, M, p["g4"], out, C2, residual=conv, ldr=C2)
            x.append(out)
            if keep is not None:
                keep[f"sa{l}_module.conv"], keep[f"sa{l}_module.out"] = conv, out
        # GlobalSAModule (model.py:134-140)
        F3, M3, B = 16 * Cw, lv[3].n, geo.B
        cat = new(M3, F3 + 4)
        self._call("concat_xyz", L.p2w_concat_xyz, ptr(x[3]), F3, ptr(lv[3].xyzr), M3, ptr(cat), F3 + 4)
        h1, h2 = new(M3, F3), new(M3, F3)
        self._gemm("gemm_mlp", cat, F3 + 4, M3, w.sa4[0], h1, F3)
        self._gemm("gemm_mlp", h1, F3, M3, w.sa4[1], h2, F3)
        g = new(B, F3)
        self._call("segment_max", L.p2w_segment_max, ptr(h2), F3, F3, ptr(lv[3].ptr), B, ptr(g))
        if keep is not None:
            keep["sa4_module.out"] = g
        # FPModule 4..1 (model.py:148-153)
        nbr4 = torch.empty(M3, dtype=torch.int32, device=dev)
        deg4 = torch.empty(M3, dtype=torch.int32, device=dev)
        self._call("fill_batch_nbr", L.p2w_fill_batch_nbr, ptr(lv[3].batch), M3, ptr(nbr4), ptr(deg4))
        zeros_c = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        y, y_xyzr = g, zeros_c
        for fl in (4, 3, 2, 1):
            fine = lv[fl - 1]
            m, Fc, Fs = fine.n, y.shape[1], x[fl - 1].shape[1]
            nbr, deg, kw = (nbr4, deg4, 1) if fl == 4 else (*geo.fp_nbr[fl - 1], 2)
            cat = new(m, Fc + Fs)
            self._call("interp_concat", L.p2w_interp_concat, ptr(y), Fc, ptr(y_xyzr), ptr(fine.xyzr), ptr(nbr), ptr(deg),
                       kw, ptr(x[fl - 1]), Fs, m, ptr(cat), Fc + Fs)
            l0, l1 = w.fp[fl]
            a, b = new(m, l0.N), new(m, l1.N)
            self._gemm("gemm_mlp", cat, Fc + Fs, m, l0, a, l0.N)
            self._gemm("gemm_mlp", a, l0.N, m, l1, b, l1.N)
            y, y_xyzr = b, fine.xyzr
            if keep is not None:
                keep[f"fp{fl}_module.out"] = b
        # head (model.py:241-243)
        hd = new(N, F3)
        self._gemm("gemm_mlp", y, F3, N, w.head1, hd, F3)
        if w.num_classes == 1:
            logits = torch.empty(N, dtype=torch.float32, device=dev)
            self._call("rowdot", L.p2w_rowdot, ptr(hd), F3, F3, ptr(w.head2_w), float(w.head2_b[0]), N, ptr(logits))
        else:
            o = new(N, w.num_classes)
            self._gemm("gemm_mlp", hd, F3, N, w.head2, o, w.num_classes)
            logits = o.t()
        return torch.squeeze(logits)

    def forward(self, pos, reflectance, ptr0, sf, keep=None):
        """One forward = the call the reference makes (``outputs = model(data)``, predicter.py:198).  With ``overlap`` (default) the
        six neighbour searches run on a second HIP stream beside the feature phase: the sampling chain (0.2 ms) comes first, the
        level sizes start their way to the host, and while the host waits for them the GPU already runs the stem, SA1's hoisted
        product and the first searches; every feature kernel that reads a search result waits for that search's event.  Results
        are those of the sequential order, bit for bit."""
        overlap = bool(self.overlap) and self.events is None
        search_stream = self._search_stream() if overlap else None
        geo = self._geometry_async(pos, reflectance, ptr0, sf, search_stream=search_stream, defer_searches=overlap)
        early = None
        if not self.early_first and geo.launch_searches is not None:
            geo.launch_searches()
        if overlap and self.prec is not None:
            # needs N only: enqueued BEFORE the searches (whose enqueue costs the host 0.15 ms) and before the host's wait for the
            # level sizes - the caller's stream goes from the sampling chain straight into the stem
            self._range_begin(sf.device)
            early = self._features_early(geo)
        if geo.launch_searches is not None:
            # search_stagger: the ball query and the level-2 kNN now; the level-3 kNN and the interpolation searches are enqueued by
            # the feature phase where it wants them to start (each behind the PointNetConv of the level before)
            geo.launch_searches(2 if (overlap and self.search_stagger and self.prec is not None) else None)
        self._geometry_finish(geo)
        if early is not None and early["lv0"] is not geo.levels[0]:   # (the table overflowed and the geometry was redone: level 0 is new)
            early = None
        if keep is not None:
            keep["geometry"] = geo
            if keep.get("geometry_only"):   # profiling: the level sizes are wanted, the fp32 copies of the level features are not
                keep = None
        geo.early = early
        self._lone = self.events is None
        try:
            logits = self.features(geo, keep)
        finally:
            self._lone = False
        if geo.search_stream is not None:     # nothing of this forward is left on the side stream when the caller gets its logits
            torch.cuda.current_stream().wait_event(geo.done)
        if geo.watch is not None:   # the range guard needs the finished phase: one more host wait per forward (Net.stream hides it)
            torch.cuda.current_stream().synchronize()
            logits = self.checked(geo, logits, keep)
        return logits

    def _search_stream(self):
        """The side stream of the single-call forward's searches (one per caller stream)."""
        pool = self.__dict__.setdefault("_search_streams", {})
        key = _lib.stream()
        if key not in pool:
            pool[key] = torch.cuda.Stream(priority=int(self.search_priority))
        return pool[key]

    # -- two-stream software pipeline over a sequence of batches ---------------------------------------------
    def forward_stream(self, inputs):
        """``inputs`` yields (pos, reflectance, ptr0, sf); yields logits per batch, in order.

        The geometry phase (VALU-bound searches) of batch i+1 runs on one HIP stream while the feature phase
        (MFMA-bound GEMMs) of batch i runs on another, high-priority one; the caller's stream only waits for the
        results.  The only host wait per batch is for the three level sizes of the NEXT batch's geometry, which has
        been running concurrently."""
        cur_stream = torch.cuda.current_stream()
        if getattr(self, "_s_geo", None) is None:
            # The feature phase is the critical path (9 of 10 ms) and its MFMA kernels leave no room on a CU for anything
            # else, so IT gets the high-priority stream and the geometry of the next batch fills what is left (next to
            # the PointNetConv kernels, at kernel tails): measured 10.09 vs 10.23 ms per step against the opposite
            # assignment, 10.26 with both high, 10.23 with both normal.
            self._s_geo = torch.cuda.Stream(priority=int(self.geo_priority))
            self._s_feat = [torch.cuda.Stream(priority=int(self.feat_priority))
                            for _ in range(max(1, self.feature_streams))]
        s_geo = self._s_geo
        f_streams = self._s_feat[: max(1, self.feature_streams)]
        yield from self._forward_stream(inputs, cur_stream, s_geo, f_streams)

    def _forward_stream(self, inputs, cur_stream, s_geo, f_streams):

        def launch_geometry(args):
            s_geo.wait_stream(cur_stream)
            for t in args:
                # the inputs were allocated on the caller's stream (possibly by a just-issued H2D copy or a dtype
                # conversion in Net._inputs) and are read on s_geo: the allocator must not recycle them under it
                t.record_stream(s_geo)
            with torch.cuda.stream(s_geo):
                geo = self._geometry_async(*args)
            geo.aux = list(geo.aux) + list(args)   # ... and they stay alive until this batch's features were launched
            return geo

        def finish(logits, ev, g, fs):
            """Hand a batch's logits to the caller's stream; with the range guard, after the host has seen the phase finish and
            its range report (the next batch's phase is already queued behind it, so the GPU does not idle meanwhile)."""
            if g.watch is not None:
                if ev is not None:
                    ev.synchronize()
                else:
                    cur_stream.synchronize()
                if self.range_violations(g.watch):
                    with torch.cuda.stream(fs):
                        logits = self.checked(g, logits, None)
                        if ev is not None:
                            ev = torch.cuda.Event()
                            ev.record()
                    if ev is not None:
                        logits.record_stream(cur_stream)
            if ev is not None:
                cur_stream.wait_event(ev)
            return logits

        it = iter(inputs)
        nxt = next(it, None)
        if nxt is None:
            return
        geo = launch_geometry(nxt)
        pending = []   # (logits, done event, geometry, feature stream) of batches whose features are in flight, oldest first
        i = 0
        while geo is not None:
            nxt = next(it, None)
            self._geometry_finish(geo)                    # host waits for THIS batch's level sizes only
            geo_next = launch_geometry(nxt) if nxt is not None else None
            fs = f_streams[i % len(f_streams)]
            fs.wait_event(geo.done)
            for t in geo.tensors():
                t.record_stream(fs)                       # allocated on s_geo, consumed on a feature stream
            if fs is cur_stream:
                logits = self.features(geo, None)
                ev = None
            else:
                fs.wait_stream(cur_stream)
                with torch.cuda.stream(fs):
                    logits = self.features(geo, None)
                    ev = torch.cuda.Event()
                    ev.record()
                logits.record_stream(cur_stream)
            pending.append((logits, ev, geo, fs))
            if len(pending) >= len(f_streams):
                yield finish(*pending.pop(0))
            geo = geo_next
            i += 1
        for item in pending:
            yield finish(*item)
