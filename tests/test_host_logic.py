"""CPU suite: the product's host logic against reference-generated golden vectors, and the
C-ABI library's export table (no compute calls without a GPU)."""
import ctypes
import json
import os
import re

import numpy as np
import pandas as pd
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(G.rstrip("/").rsplit("/", 1)[0])


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_label_sets_and_constants():
    from lemon_amd import datasets as ds
    meta = json.load(open(os.path.join(G, "dataset_meta.json")))
    assert ds.cifar10_labels.tolist() == meta["labels"]["cifar10"] and len(ds.cifar10_labels) == 10
    assert ds.cifar100_labels.tolist() == meta["labels"]["cifar100"] and len(ds.cifar100_labels) == 100
    assert len(ds.mini_imagenet_labels) == meta["class_num_dict"]["mini_imagenet"] == 100
    assert len(ds.stanford_cars_labels) == meta["class_num_dict"]["stanford_cars"] == 196
    assert ds.CLIP_MEAN == meta["CLIP_MEAN"] and ds.CLIP_STD == meta["CLIP_STD"]


@pytest.mark.parametrize("dataset", ["cifar10", "cifar100"])
def test_label_noise_streams_match_reference(dataset):
    from lemon_amd import datasets as ds
    g = load("noise_labels.npz")
    y = g[f"{dataset}_y"]
    for seed in (0, 1, 2):
        for lvl in (0.2, 0.4):
            for kind in ("asymmetric", "symmetric"):
                got = ds.add_noisy_labels(dataset, kind, lvl, seed, list(y))
                assert np.array_equal(got, g[f"{dataset}_{kind}_{seed}_{lvl}"]), (kind, seed, lvl)
    got = ds.add_noisy_labels(dataset, "asymmetric", 0.4, 0, list(y))
    assert 0.35 < (got != y).mean() < 0.45
    assert set(np.unique((got - y) % ds.class_num_dict[dataset])) == {0, 1}     # pair flip: c -> c+1


def test_cat_noise_on_cifar_raises_like_the_reference():
    from lemon_amd import datasets as ds
    assert int(load("noise_labels.npz")["cat_raises"]) == 1
    with pytest.raises(NotImplementedError):
        ds.add_noisy_labels("cifar100", "cat", 0.4, 0, [0, 1, 2])


def test_split_indices_match_reference():
    import hashlib
    from lemon_amd import datasets as ds
    g = load("splits.npz")
    for seed in (0, 1, 2):
        tr, va, te = ds.split_80_10_10(50000, seed)
        assert (len(tr), len(va), len(te)) == (40000, 5000, 5000)
        shas = [hashlib.sha256(np.ascontiguousarray(a.astype(np.int64)).tobytes()).hexdigest() for a in (tr, va, te)]
        assert shas == g[f"sha_{seed}"].tolist()
        assert np.array_equal(np.stack([tr[:16], va[:16], te[:16]]), g[f"head_{seed}"])
    tr, va, te = ds.split_80_10_10(50000, 0)
    assert np.array_equal(tr, g["train_0"]) and np.array_equal(va, g["val_0"]) and np.array_equal(te, g["test_0"])


def test_caption_noise_matches_reference():
    from lemon_amd import datasets as ds
    g = load("noise_captioning.npz")
    for seed in (0, 7):
        d = ds.random_noise_dict(50, 0.4, seed)
        assert np.array_equal(np.array(list(d.keys())), g[f"random_{seed}_keys"])
        assert np.array_equal(np.array(list(d.values())), g[f"random_{seed}_vals"])
        assert all(k != v for k, v in d.items())
    flat, lens = g["cats_flat"], g["cats_len"]
    cats, p = [], 0
    for n in lens:
        cats.append(flat[p:p + n].tolist()); p += n
    for seed in (0, 3):
        d = ds.calc_noise_by_integer_matching(np.array(cats, dtype=object), 0.4, seed)
        assert np.array_equal(np.array(list(d.keys())), g[f"match_{seed}_keys"])
        assert np.array_equal(np.array(list(d.values())), g[f"match_{seed}_vals"])
    frame = pd.DataFrame({"sentence": [f"caption {i % 37}" for i in range(60)]}, index=np.arange(100, 160))
    noised = ds.noise_given_dict(frame, ds.random_noise_dict(60, 0.3, 1))
    assert np.array_equal(np.array([int(s.split()[1]) for s in noised["sentence"]]), g["given_sentence_id"])
    assert np.array_equal(noised["is_mislabel"].values.astype(np.uint8), g["given_is_mislabel"])


# ------------------------------------------------------------------ the C-ABI boundary
def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lemon_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lemon_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from lemon_amd import _lib
    names = _declared_symbols()
    assert sorted(_lib.EXPORTS) == names, "the ctypes binding and include/lemon_hip.h disagree"
    assert os.path.exists(_lib.SO_PATH), "liblemon_hip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/lemon_hip.h but not exported"
    # signatures are plain C: no torch / C++ types leak into the ABI
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "lemon_hip.h")).read(), flags=re.S)
    assert "torch" not in hdr and "at::" not in hdr and "std::" not in hdr and "hipStream_t" not in hdr


def test_library_contains_gfx950_code_object():
    from lemon_amd import _lib
    blob = open(_lib.SO_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_scan_f32" in blob


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under lemon_amd/ or include/ may reference it."""
    bad = []
    for base in ("lemon_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="replace").read()
                    if re.search(r"\boracle\b", txt) and "oracle" in txt.replace("CPU oracle", "").replace("the oracle", "").replace("oracle's", ""):
                        if re.search(r"(import|from|include|CDLL|dlopen).*oracle", txt):
                            bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_host_inputs_are_refused_without_gpu_fallback():
    import torch
    from lemon_amd import ops, _lib
    with pytest.raises((_lib.LemonHipError, TypeError)):
        ops.normalize_vectors(torch.zeros(3, 4))        # CPU tensor: no silent CPU path
    with pytest.raises(TypeError):
        ops.normalize_vectors(np.zeros((3, 4), np.float32))


def test_linear_results_file_is_stamped_and_well_formed():
    # lemon_amd/data/linear_gfx950.csv: recorded hipBLASLt solutions (written by tools/tune_gemms.py on an MI355X).  The
    # stamp line ties it to a hipBLASLt version + arch; lemon_linear_load_tuned ignores the file on a mismatch.
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lemon_amd", "data", "linear_gfx950.csv")
    lines = [l.strip() for l in open(path) if l.strip()]
    assert re.fullmatch(r"# lemon_linear hipblaslt=\d+ arch=gfx950", lines[0]), lines[0]
    rows = [l.split(",") for l in lines if not l.startswith("#")]
    assert rows and all(len(r) in (7, 8) for r in rows)            # 8th column: operand type (1 = bf16 split operands, k = 6 x width)
    keys = {tuple(int(v) for v in r[:5]) + (int(r[7]) if len(r) == 8 else 0,) for r in rows}
    assert len(keys) == len(rows)                                   # one solution per (m,n,k,epilogue,residual,operand type)
    assert any(k[1:3] == (2304, 768) and k[5] == 0 for k in keys)   # ViT-B/32 QKV projection, fp32 GEMM
    assert any(k[1:3] == (2304, 6 * 768) and k[5] == 1 for k in keys)   # ... and as the 3-way bf16 split GEMM


def test_maximize_metric_survives_a_diverging_lbfgs_candidate():
    """On embeddings WITHOUT CLIP's modality gap (paired image-text cosine ~0.9, so dists_tr ~0.1 and D = -cos ~ -0.9) the
    SoftMargin proxy of lib/metrics/utils.py:121-149 overflows from the start point [10]*6: LBFGS returns NaN and the
    reference itself dies in fminbound ("Optimization bounds must be finite scalars" -- observed when
    tools/make_golden_loop.py first ran the reference on such data).  Our search skips the non-finite candidate
    (metrics.py, documented divergence) and still returns the grid / scipy optimum."""
    import torch
    from lemon_amd import metrics as M
    rs = np.random.RandomState(0)
    n, k = 120, 5
    y = (rs.rand(n) < 0.4).astype(np.int64)
    rec = {"d_1": (0.1 + 0.05 * rs.rand(n) + 0.05 * y).astype(np.float64),
           "D_n": -(0.85 + 0.1 * rs.rand(n, k)).astype(np.float32), "dists_tr_n": (0.05 + 0.1 * rs.rand(n, k)).astype(np.float32),
           "dists_n": (0.3 * rs.rand(n, k) + 0.3 * y[:, None]).astype(np.float32),
           "D_m": -(0.85 + 0.1 * rs.rand(n, k)).astype(np.float32), "dists_tr_m": (0.05 + 0.1 * rs.rand(n, k)).astype(np.float32),
           "dists_m": (0.3 * rs.rand(n, k) + 0.3 * y[:, None]).astype(np.float32)}
    rec_t = {k_: torch.as_tensor(v, dtype=torch.float64 if k_ == "d_1" else torch.float32) for k_, v in rec.items()}
    cand = M._torch_lbfgs(rec_t, y, [10.0] * 6, (), ())
    assert not np.all(np.isfinite(cand))                            # the proxy really diverges on this frame

    def score_fn(hp):
        sn = np.exp(-hp["tau_1_n"] * rec["D_n"]) * np.exp(-hp["tau_2_n"] * rec["dists_tr_n"])
        sm = np.exp(-hp["tau_1_m"] * rec["D_m"]) * np.exp(-hp["tau_2_m"] * rec["dists_tr_m"])
        return rec["d_1"] + hp["beta"] * (sn * rec["dists_n"]).mean(1) + hp["gamma"] * (sm * rec["dists_m"]).mean(1)

    grid = {"beta": [0, 5], "gamma": [0, 5], "tau_1": [0, 1], "tau_2": [0, 5]}
    best_x, best_val, thres = M.maximize_metric(score_fn, y, grid, [[0] * 6, [10] * 6], M.optimize_f1_efficient, {},
                                                scipy_methods=("Nelder-Mead",), rec_for_lbfgs=rec)
    assert np.all(np.isfinite(best_x)) and np.isfinite(thres) and 0.5 < best_val <= 1.0


@pytest.mark.parametrize("h,w,oh,ow", [(32, 32, 224, 224), (48, 64, 224, 298), (300, 200, 336, 224), (500, 375, 298, 224),
                                       (37, 91, 224, 550), (640, 480, 298, 224)])
def test_pil_bicubic_tables_reproduce_pil_resize(h, w, oh, ow):
    # lemon_amd/data.py::pil_bicubic_tables (what lemon_preprocess_u8 consumes) against PIL itself
    from PIL import Image
    from lemon_amd.data import PIL_PRECISION_BITS, pil_bicubic_tables

    def axis(img, out_size):          # resample axis 1 of a uint8 [H, W, C] array
        kk, b = pil_bicubic_tables(img.shape[1], out_size)
        out = np.zeros((img.shape[0], out_size, img.shape[2]), np.uint8)
        for xx in range(out_size):
            x0, n = b[xx]
            acc = (1 << (PIL_PRECISION_BITS - 1)) + (img[:, x0:x0 + n].astype(np.int64) * kk[xx, :n][None, :, None]).sum(1)
            out[:, xx] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        return out

    img = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = axis(axis(img, ow).transpose(1, 0, 2), oh).transpose(1, 0, 2)      # horizontal pass, then vertical
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(got, ref)


def test_embedding_cache_roundtrip_and_keying(tmp_path):
    """lemon_amd/cache.py: an entry is found again only for the same dataset / noise / checkpoint / shard / prompts, and a
    partially written entry (no `done` marker) is never read."""
    import torch
    from lemon_amd.cache import EmbeddingCache
    c = EmbeddingCache(str(tmp_path), dataset="cifar10", noise_type="asymmetric", noise_level=0.4, data_seed=0, clip_path="random")
    img, txt = torch.randn(5, 8), torch.randn(5, 8)
    prompts = [f"A photo of a thing{i}" for i in range(5)]
    meta = {"prompts": prompts, "lo": 0}
    assert c.load("train", 0, 5, prompts, "cpu") is None
    c.store("train", 0, 5, prompts, img, txt, meta)
    hit = c.load("train", 0, 5, prompts, "cpu")
    assert hit is not None and torch.equal(hit[0], img) and torch.equal(hit[1], txt) and hit[2]["prompts"] == prompts
    assert c.load("val", 0, 5, prompts, "cpu") is None                         # other split
    assert c.load("train", 0, 5, prompts[::-1], "cpu") is None                 # other noise realisation
    other = EmbeddingCache(str(tmp_path), dataset="cifar10", noise_type="asymmetric", noise_level=0.2, data_seed=0, clip_path="random")
    assert other.load("train", 0, 5, prompts, "cpu") is None
    entry = [d for d in tmp_path.iterdir() if d.is_dir()][0]
    (entry / "done").unlink()
    assert c.load("train", 0, 5, prompts, "cpu") is None
    assert EmbeddingCache(None).load("train", 0, 5, prompts, "cpu") is None    # disabled cache


@pytest.mark.parametrize("panels,tiles", [(391, 313), (2048, 2048), (40, 313), (1, 1), (1, 313), (7, 5), (725, 391),
                                          (157, 7813), (64, 10), (3, 2000), (8192, 3), (513, 129)])
def test_scan_plan_covers_every_unit_once(panels, tiles):
    """lemon_plan_segments (csrc/knn_f32.hip) through the host-only lemon_debug_scan_plan: the segments of all workgroups tile
    the panels x tiles unit space exactly once, stay inside their panel, number a panel's pieces 0 .. pieces-1 without gaps
    (k_merge reads exactly those slots), and no workgroup carries much more than its share."""
    import ctypes
    from lemon_amd import _lib
    lib = _lib.load()
    cap_wgs, cap_segs = 4096, panels * 8 + 8192
    g, sp, ns = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    sb = np.zeros(cap_wgs + 1, np.int32); pc = np.zeros(panels, np.int32); sg = np.zeros(4 * cap_segs, np.int32)
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    rc = lib.lemon_debug_scan_plan(panels, tiles, ctypes.byref(g), ctypes.byref(sp), ip(sb), cap_wgs, ip(pc), ip(sg), cap_segs,
                                   ctypes.byref(ns))
    assert rc == 0
    grid, segs = g.value, sg[:4 * ns.value].reshape(-1, 4)
    assert sb[0] == 0 and sb[grid] == ns.value and (np.diff(sb[:grid + 1]) >= 0).all()
    cover = np.zeros((panels, tiles), np.int32)
    seen = [set() for _ in range(panels)]
    for pnl, t0, nt, piece in segs:
        assert 0 <= pnl < panels and t0 >= 0 and nt >= 1 and t0 + nt <= tiles
        assert 0 <= piece < pc[pnl] and piece not in seen[pnl]
        seen[pnl].add(piece)
        cover[pnl, t0:t0 + nt] += 1
    assert (cover == 1).all()
    assert all(len(seen[p_]) == pc[p_] for p_ in range(panels)) and sp.value == pc.max()
    work = np.array([segs[sb[b]:sb[b + 1], 2].sum() + 3 * (sb[b + 1] - sb[b]) for b in range(grid)])   # tiles + 3 per segment
    share = (panels * tiles + 3 * len(segs)) / grid
    assert work.max() <= 1.25 * share + tiles * 0 + 8, (work.max(), share)


def test_caption_length_bucketing_is_exact_and_cuts_batches_at_bucket_boundaries():
    # pipeline.Embedder(length_bucketing=True): captions sorted by length, micro-batches cut at the text tower's 8-token bucket
    # boundaries, every caption run at its own bucket's token count -- exact under the causal mask (run_lemon.py:140-154 pads every
    # caption to the full context instead).  CPU, tiny model: same embeddings as the plain order, in the caller's order.
    import torch
    from lemon_amd.clip import ClipConfig, LemonCLIP
    from lemon_amd.pipeline import Embedder
    cfg = ClipConfig.named("tiny")
    model = LemonCLIP(cfg).eval()
    g = torch.Generator().manual_seed(3)
    n = 37
    length = torch.randint(3, cfg.context_length + 1, (n,), generator=g)
    ids = torch.zeros((n, cfg.context_length), dtype=torch.long)
    for i, L in enumerate(length.tolist()):
        ids[i, :L - 1] = torch.randint(1, cfg.eos_token_id, (L - 1,), generator=g)
        ids[i, L - 1] = cfg.eos_token_id
    plain = Embedder(model, torch.device("cpu"), batch_size=4, text_batch_size=5)
    bucketed = Embedder(model, torch.device("cpu"), batch_size=4, text_batch_size=5, length_bucketing=True)
    calls = []
    real = model.encode_text
    model.encode_text = lambda x, *a, **k: (calls.append((int(x.shape[0]), k.get("seq_len"))), real(x, *a, **k))[1]
    try:
        eot = ids.argmax(-1)                                   # (the un-normalised embeddings: normalize_vectors is a GPU kernel)
        e0 = plain._embed_texts(ids, eot)
        calls.clear()
        e1 = bucketed._embed_texts(ids, eot)
    finally:
        model.encode_text = real
    assert float((e0 - e1).abs().max()) < 2e-6
    # every micro-batch holds captions of ONE bucket, at most text_batch_size of them, and runs exactly that bucket's tokens
    assert sum(c[0] for c in calls) == n and all(c[0] <= 5 for c in calls)
    buckets = sorted(model.text.seq_len_for(int(L) - 1) for L in length.tolist())
    ran = sorted(L for cnt, L in calls for _ in range(cnt))
    assert ran == buckets
    assert bucketed.text_tokens_run == sum(buckets) and plain.text_tokens_run >= bucketed.text_tokens_run
    # text_token_budget: a micro-batch holds budget // L captions of its bucket (same GEMM row count for short and long captions)
    budgeted = Embedder(model, torch.device("cpu"), batch_size=4, text_batch_size=5, length_bucketing=True, text_token_budget=48)
    calls.clear()
    model.encode_text = lambda x, *a, **k: (calls.append((int(x.shape[0]), k.get("seq_len"))), real(x, *a, **k))[1]
    try:
        e2 = budgeted._embed_texts(ids, eot)
    finally:
        model.encode_text = real
    assert float((e0 - e2).abs().max()) < 2e-6 and sum(c[0] for c in calls) == n
    assert all(cnt <= 48 // L for cnt, L in calls) and any(cnt > 5 for cnt, L in calls if L == 8)


def test_near_tie_adjudication_names_the_side_that_holds_the_float64_set():
    # oracle/reference_loop.adjudicate_near_ties (bench.py's cpu_baseline leg, tests/test_gpu_parity.py): two searches that differ
    # in one row; the float64 top-k over the whole DB decides, the gap is the float64 distance of the rows in dispute
    import numpy as np
    from oracle import reference_loop as rl
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3000, 64)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = x[:200].copy()
    s = q.astype(np.float64) @ x.astype(np.float64).T
    order = np.argsort(-s, 1)
    I = order[:, :10].copy()
    J = I.copy()
    J[5, 9] = order[5, 10]                                   # row 5 of the second search holds the 11th best instead of the 10th
    rep = rl.adjudicate_near_ties(q, x, I, J)
    assert rep["rows_differing"] == 1 and rep["worst_row"] == 5
    assert rep["rows_a_equals_f64_set"] == 1.0 and abs(rep["rows_b_equals_f64_set"] - 199 / 200) < 1e-12
    assert abs(rep["max_gap_at_swap"] - (s[5, order[5, 9]] - s[5, order[5, 10]])) < 1e-12
    # train queries: the self match is left out of the float64 ranking
    rep = rl.adjudicate_near_ties(q, x, order[:, 1:11], order[:, 1:11], exclude=np.arange(200))
    assert rep["rows_differing"] == 0 and rep["rows_a_equals_f64_set"] == 1.0
