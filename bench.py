#!/usr/bin/env python3
"""bench.py -- LEMoN hot path (CLIP embed -> brute-force kNN -> label-error scores) on MI355X.

  python bench.py --gpus N --steps K --warmup W
(for N > 1 the driver launches it under torch.distributed.run, one rank per GPU over RCCL).

A step = one pass of the hot path over one batch of synthetic input already resident in HBM.
Default workload = BASELINE.json configs[1]: CIFAR-100 shape, ViT-B/32, 40 000 train (= DB) +
5 000 val + 5 000 test samples PER GPU (weak scaling: with N GPUs the DB is the all-gathered
N x 40 000 rows), 40 % pair-flip label noise, cosine distance, k = 50.  Rank 0 prints ONE JSON line.

Other workloads (not the headline line; used for profiles and DESIGN.md numbers):
  --workload knn      synthetic N x d unit embeddings, self-join with self-exclusion (configs[3] shape,
                      default 1M x 768, k=50): kNN + scoring only, no encoder.
  --workload mscoco   BASELINE configs[2] shape: ViT-B/16, 82 783 + 5 000 + 5 000 image / caption pairs, every caption
                      distinct and 8 ... 77 tokens long, DB = a random 50 000-row subset of the train split
                      (run_lemon.py:121-127), one GPU.  The N = 1 headline line carries a bounded-size run of it
                      (`mscoco`: 8 000 + 500 + 500 samples).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters (dense fp32 matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak (6.3 TB/s achievable)


def pmc_summary(kernel_substr, fname="bench_default_pmc.csv"):
    """{counter: row} of the longest-running launch group of a kernel in a COMMITTED PMC summary (profiles/rN/<fname>), + path."""
    import csv
    path = next((p for p in (os.path.join(ROOT, "profiles", r, fname) for r in ("r5", "r4", "r3", "r2", "r1")) if os.path.exists(p)), None)
    if path is None:
        return {}, None
    best = {}
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["kernel"] and int(r["grid"]) >= 256 * 256:
            # (steady-state group: the most launches, then the longest)
            key = (int(r["launches"]), float(r["dur_ms"]))
            if r["counter"] not in best or key > best[r["counter"]][0]:
                best[r["counter"]] = (key, r)
    return {c: v[1] for c, v in best.items()}, os.path.relpath(path, ROOT)


def pmc_traffic(kernel_substr):
    """(bytes, source): HBM-side bytes per launch of the dominant kernel from the COMMITTED rocprofv3 --pmc passes of
    this command (tools/gpu_profile.sh; FETCH_SIZE and WRITE_SIZE in separate runs, KB units, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for 16-B-per-lane streaming reads on gfx950).  PMC counters cannot be collected
    from inside the timed process, so this is NOT a measurement of the current run: the file it came from is
    reported next to it as `traffic_source`.  (None, None) when no summary is committed."""
    import csv
    path = next((p for p in (os.path.join(ROOT, "profiles", r, "bench_default_pmc.csv") for r in ("r5", "r4", "r3", "r2", "r1"))
                 if os.path.exists(p)), None)
    if path is None:
        return None, None
    # the dominant launch = the longest-running group of this kernel (the summary keeps launches of different problems apart)
    best = {}
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["kernel"] and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE") and int(r["grid"]) >= 256 * 256:
            if r["counter"] not in best or float(r["dur_ms"]) > float(best[r["counter"]]["dur_ms"]):
                best[r["counter"]] = r
    if len(best) < 2:
        return None, None
    return (2.0 * float(best["FETCH_SIZE"]["value_KB"]) + float(best["WRITE_SIZE"]["value_KB"])) * 1024.0, os.path.relpath(path, ROOT)


def hbm_scan_model_object(panel, achieved_gbs):
    """SURVEY 8d's scan-model bytes (one DB stream per `panel`-query panel) over the kernel time.  The kernels are MFMA-bound
    and keep the DB tiles in LDS / L2 across many more queries than one panel, so the MODEL rate can exceed the HBM peak: it is
    then reported as `model_over_peak` (a statement about the model, not a bandwidth the chip delivered) and `frac` is left out."""
    o = {"B": panel, "achieved_GBs": achieved_gbs, "peak_GBs": PEAK_HBM_GBS,
         "note": "MODEL bytes, not measured traffic; the binding fraction is the MFMA one"}
    if achieved_gbs <= PEAK_HBM_GBS:
        o["frac"] = achieved_gbs / PEAK_HBM_GBS
    else:
        o["model_over_peak"] = achieved_gbs / PEAK_HBM_GBS
    return o


hbm_model_object = hbm_scan_model_object


def pmc_gemm_summary():
    """({traffic, mfma_busy, clock_GHz}, source) of the hand-written GEMM's longest-running launch group in the committed PMC
    summary of the headline command (profiles/rN/encoder_gemm_pmc.csv, tools/gpu_profile.sh): FETCH_SIZE x 2 + WRITE_SIZE bytes
    per launch, MFMA-busy fraction and shader clock.  Not a measurement of the current run."""
    import csv
    path = next((p for p in (os.path.join(ROOT, "profiles", r, "encoder_gemm_pmc.csv") for r in ("r5", "r4")) if os.path.exists(p)), None)
    if path is None:
        return {}, None
    best = {}
    for r in csv.DictReader(open(path)):
        if "k_gemm_f16x3t" in r["kernel"]:
            key = float(r["dur_ms"]) * int(r["launches"])
            if r["counter"] not in best or key > best[r["counter"]][0]:
                best[r["counter"]] = (key, r)
    g = {c: v[1] for c, v in best.items()}
    out = {}
    if "FETCH_SIZE" in g and "WRITE_SIZE" in g:
        out["traffic"] = (2.0 * float(g["FETCH_SIZE"]["value_KB"]) + float(g["WRITE_SIZE"]["value_KB"])) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
        act = float(g["GRBM_GUI_ACTIVE"]["value_KB"])
        out["mfma_busy"] = float(g["SQ_VALU_MFMA_BUSY_CYCLES"]["value_KB"]) / (act / 8.0 * 1024.0)
        out["clock_GHz"] = act / 8.0 / (float(g["GRBM_GUI_ACTIVE"]["dur_ms"]) * 1e6)
    return out, os.path.relpath(path, ROOT)


def sustained_bf16_peak():
    """(TFLOP/s, source): what bf16 MFMAs on RANDOM register operands sustain on this board with no memory traffic at all
    (tools/micro/mfma_peak.hip; the board's power management lowers the shader clock), read from the COMMITTED output of
    that micro-benchmark -- context for the bf16 scan's `frac` of the data-sheet peak, not a measurement of this run."""
    import re
    path = next((p for p in (os.path.join(ROOT, "profiles", r, "micro_mfma_peak.txt") for r in ("r5", "r4", "r3", "r2")) if os.path.exists(p)), None)
    if path is None:
        return None, None
    for ln in open(path):
        m = re.search(r"bf16 32x32x16 sustained, random\s+operands.*?([0-9.]+) TFLOP/s", ln)
        if m:
            return float(m.group(1)), os.path.relpath(path, ROOT)
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cifar100", choices=["cifar100", "knn", "mscoco"])
    ap.add_argument("--arch", default=None, help="vit-b-32 (cifar100 default), vit-b-16 (mscoco default), vit-l-14")
    ap.add_argument("--n_train", type=int, default=None, help="default 40 000 (cifar100) / 82 783 (mscoco)")
    ap.add_argument("--n_val", type=int, default=5000)
    ap.add_argument("--n_test", type=int, default=5000)
    ap.add_argument("--db_limit", type=int, default=50000,
                    help="mscoco: --compr_dataset_size_limit of run_lemon.py:48 (DB = a random subset of the train split, :121-127)")
    ap.add_argument("--image_hw", default="256x320", help="mscoco: height x width of the decoded uint8 images resident in HBM")
    ap.add_argument("--no_length_bucketing", action="store_true", help="mscoco: captions not sorted by length (every micro-batch runs ~77 tokens)")
    ap.add_argument("--knn_k", type=int, default=50)
    ap.add_argument("--dist_type", default="cosine", choices=["cosine", "euclidean"])
    ap.add_argument("--encoder_batch", type=int, default=5240,
                    help="images per encoder micro-batch (per-sample results are equal within fp32 rounding whatever it is: the hand-written GEMM is position-independent, "
                         "a library GEMM may pick another solution -- another summation order -- for another row count); 5 240 x 50 tokens = 2 048 row tiles "
                         "of the block GEMMs: whole rounds of the 512 workgroup slots (2 620 = 1 024 tiles measured 0.8 %% slower on the same box, 10 480 and "
                         "20 000 the same as 5 240: half as many launches and tile-tail rounds per image)")
    ap.add_argument("--text_dedup", action="store_true",
                    help="embed each distinct prompt once (exact; off by default so every sample's prompt is encoded)")
    ap.add_argument("--algo", default="auto", choices=["auto", "f32", "bf16"])
    ap.add_argument("--knn_n", type=int, default=1_000_000)
    ap.add_argument("--knn_d", type=int, default=768)
    ap.add_argument("--input", default="u8", choices=["u8", "f32"],
                    help="u8: raw 32x32 uint8 images resident in HBM, preprocessing (bicubic resize to 224, normalise) "
                         "inside the timed step; f32: already preprocessed pixel tensors")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_f32_gemm_check", action="store_true", help="skip the extra untimed step with every GEMM in fp32")
    ap.add_argument("--cpu_sample_images", type=int, default=96)
    ap.add_argument("--cpu_sample_queries", type=int, default=2048)
    ap.add_argument("--no_knn_1m", action="store_true",
                    help="skip the extra 1M x 768 self-join object (`knn_1m`) the N=1 headline line carries")
    ap.add_argument("--no_mscoco", action="store_true",
                    help="skip the bounded caption-shaped run (`mscoco`: BASELINE configs[2] at 8 000 + 500 + 500 samples) the N=1 headline line carries")
    ap.add_argument("--rccl_world1", action="store_true",
                    help="N = 1 only: initialise a one-rank RCCL process group and send the DB all-gathers through ncclAllGather anyway "
                         "(the line then carries `exchange`)")
    ap.add_argument("--knn_cpu_queries", type=int, default=10000,
                    help="query slice of the kNN workload's CPU baseline (BASELINE.md 3.5: 10 000 queries against the full DB)")
    args = ap.parse_args()
    if args.arch is None:
        args.arch = "vit-b-16" if args.workload == "mscoco" else "vit-b-32"
    if args.n_train is None:
        args.n_train = 82783 if args.workload == "mscoco" else 40000
    if args.workload == "mscoco" and args.encoder_batch == 5240:
        args.encoder_batch = 664                 # 664 x 197 tokens = 1 022 row tiles of 128 (ViT-B/16: see tools/gpu_profile.sh)
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` started by hand (no torchrun): start N fresh rank processes BEFORE anything touches the
    GPU in this one, wait for them, relay rank 0's JSON line, fail if any rank fails.  No exec: children are ordinary
    subprocesses with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run sets)."""
    import socket
    import subprocess
    # the rendezvous port is HANDED to rank 0 as an open listening socket (torch's TCPStore takes `master_listen_fd`): no
    # bind-close-reuse window in which another process could take the port
    lsock, port = open_rendezvous_socket()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env["LEMON_MASTER_HANDOFF"] = "1"
        if r == 0:
            env["LEMON_MASTER_LISTEN_FD"] = str(lsock.fileno())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      pass_fds=(lsock.fileno(),) if r == 0 else ()))
    lsock.close()                                # rank 0 owns the listening socket now
    # rank 0's line is read by a thread (its pipe must be drained while we poll); every child is polled, and the first
    # non-zero exit terminates the others -- a rank that dies at start-up must not leave the rest in the rendezvous
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, pr in enumerate(procs):
            if codes[i] is None:
                codes[i] = pr.poll()
        if any(c not in (None, 0) for c in codes):
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    pr.terminate()
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = pr.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[i] = pr.wait()
            break
        time.sleep(0.2)
    reader.join(timeout=5)
    sys.stdout.write(b"".join(out).decode())
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"bench.py: rank exit codes {codes}")
    return 0


def rendezvous_store(rank, world):
    """The TCPStore of this rank's process group, or None for torch's default env:// rendezvous (what torch.distributed.run
    sets up).  Under bench.py's own launcher (LEMON_MASTER_HANDOFF=1) rank 0 takes over the launcher's LISTENING socket
    (LEMON_MASTER_LISTEN_FD) and the other ranks connect to its port; a one-rank group without MASTER_PORT lets the OS pick."""
    import datetime
    import torch.distributed as dist
    fd = os.environ.pop("LEMON_MASTER_LISTEN_FD", None)
    if world == 1 and "MASTER_PORT" not in os.environ:
        return dist.TCPStore("127.0.0.1", 0, 1, is_master=True, timeout=datetime.timedelta(seconds=120))
    if os.environ.get("LEMON_MASTER_HANDOFF") != "1":
        return None
    port = int(os.environ["MASTER_PORT"])
    if rank == 0:
        if fd is not None:
            try:
                return dist.TCPStore("127.0.0.1", port, world, is_master=True, timeout=datetime.timedelta(seconds=300),
                                     master_listen_fd=int(fd), use_libuv=False)
            except Exception as e:                 # a torch without master_listen_fd: bind the port ourselves
                print(f"bench.py: listen-socket hand-off failed ({e!r}); binding MASTER_PORT directly", file=sys.stderr)
                os.close(int(fd))
        return dist.TCPStore("127.0.0.1", port, world, is_master=True, timeout=datetime.timedelta(seconds=300), use_libuv=False)
    return dist.TCPStore("127.0.0.1", port, world, is_master=False, timeout=datetime.timedelta(seconds=300), use_libuv=False)


def open_rendezvous_socket():
    """(listening socket, port) for rank 0 to inherit: bound and listening BEFORE any rank starts."""
    import socket
    lsock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    lsock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    lsock.bind(("127.0.0.1", 0))
    lsock.listen(128)
    lsock.set_inheritable(True)
    return lsock, lsock.getsockname()[1]


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                       # before any rendezvous: a mismatch must fail, not hang or silently shrink
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (torch.cuda.is_available() is False)")
    local = local % max(torch.cuda.device_count(), 1)     # rehearsals: several ranks may share one card
    torch.cuda.set_device(local)
    # --rccl_world1: a ONE-rank RCCL group, with the all-gathers forced through the collective (pipeline.all_gather_rows):
    # the exchange step of SURVEY 8e executes on the one-GPU box (ncclCommInitRank, ncclAllGather, the `exchange` report)
    if world > 1 or args.rccl_world1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LEMON_DIST_BACKEND", "nccl")    # nccl = RCCL; "gloo" only to rehearse on one card
        store = rendezvous_store(rank, world)
        kw = dict(store=store) if store is not None else {}
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        if args.rccl_world1:
            os.environ["LEMON_FORCE_ALLGATHER"] = "1"
    return world, rank, torch.device("cuda", local)


def barrier_sync(world, dev):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize(dev)


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------ workloads
def make_cifar_like(args, cfg, rank, dev):
    """Synthetic CIFAR-100-shaped input, resident in HBM: raw 32x32 uint8 images (what the CIFAR
    pickles hold; generic_transform, lib/datasets/utils.py:163-170, then runs inside the step) or, with
    --input f32, already preprocessed pixel tensors; and tokenised prompts
    'A photo of a <noisy label>' (run_lemon.py:117-119,140-146)."""
    from lemon_amd import datasets as ds
    from lemon_amd.clip import SyntheticTokenizer
    tok = SyntheticTokenizer(cfg.vocab_size, cfg.context_length, cfg.eos_token_id)
    prompts = ["A photo of a " + l for l in ds.cifar100_labels]
    class_ids = torch.tensor(tok(prompts, padding="max_length", truncation=True)["input_ids"])
    rng = np.random.default_rng(1000 + rank)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    data = {}
    for name, n in (("train", args.n_train), ("val", args.n_val), ("test", args.n_test)):
        clean = rng.integers(0, 100, n)
        flip = rng.random(n) < 0.4
        noisy = np.where(flip, (clean + 1) % 100, clean)          # pair-flip ("asymmetric") noise
        if args.input == "u8":       # raw CIFAR-shaped images: generic_transform runs inside the timed step (on the GPU)
            # planted structure: a fixed random pattern per CLEAN class + per-image noise, so that even a
            # random-init encoder puts same-class images near each other and the AUROC check is informative
            pat = torch.randint(0, 256, (100, 32, 32, 3), dtype=torch.int16, device=dev,
                                generator=torch.Generator(device=dev).manual_seed(7))
            noise = torch.randint(-96, 97, (n, 32, 32, 3), dtype=torch.int16, device=dev, generator=g)
            px = (pat[torch.from_numpy(clean).to(dev)] + noise).clamp_(0, 255).to(torch.uint8)
        else:
            px = torch.empty((n, 3, cfg.image_size, cfg.image_size), dtype=torch.float32, device=dev)
            for i in range(0, n, 2000):                           # chunked: keeps the RNG workspace small
                px[i:i + 2000].normal_(generator=g)
        data[name] = dict(pixels=px, ids=class_ids[torch.from_numpy(noisy)].to(dev),
                          label_id=torch.from_numpy(noisy.astype(np.int32)).to(dev),
                          clean=clean, noisy=noisy)
    return data


def make_mscoco_like(args, cfg, rank, dev):
    """Synthetic mscoco-shaped input (BASELINE configs[2]; lib/datasets/utils.py:275-323), resident in HBM: decoded uint8 images
    [n, H, W, 3] (generic_transform -- bicubic Resize(224, shorter side) + CenterCrop + Normalize, lib/datasets/utils.py:163-170 --
    runs inside the timed step, on the GPU) and one caption per image as token ids: BOS, 6 ... 75 random word tokens, EOT, zero
    padding to the 77-token context (what `tokenizer(texts, padding="max_length", truncation=True)` hands over, run_lemon.py:151-154)
    -- every caption distinct, lengths uniform on 8 ... 77 tokens.  Noise: 40 % of the samples carry another sample's caption.
    The DB is a random `--db_limit` subset of the train split, drawn as run_lemon.py:121-124 draws it."""
    H, W = (int(v) for v in args.image_hw.lower().split("x"))
    g = torch.Generator(device=dev).manual_seed(2000 + rank)
    rng = np.random.default_rng(2000 + rank)
    eos, bos = cfg.eos_token_id, cfg.eos_token_id - 1
    data = {}
    for name, n in (("train", args.n_train), ("val", args.n_val), ("test", args.n_test)):
        px = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
        for i in range(0, n, 4096):                                # chunked: keeps the RNG workspace small
            m = min(4096, n - i)
            base = torch.randint(0, 256, (m, H // 16, W // 16, 3), dtype=torch.uint8, device=dev, generator=g)     # blocky content + pixel noise
            px[i:i + m] = (base.repeat_interleave(16, 1).repeat_interleave(16, 2).to(torch.int16)
                           + torch.randint(-24, 25, (m, H, W, 3), dtype=torch.int16, device=dev, generator=g)).clamp_(0, 255).to(torch.uint8)
        length = rng.integers(8, cfg.context_length + 1, n)          # tokens incl. BOS and EOT
        ids = np.zeros((n, cfg.context_length), dtype=np.int64)
        words = rng.integers(1, bos, (n, cfg.context_length))
        col = np.arange(cfg.context_length)[None, :]
        ids = np.where(col < (length - 1)[:, None], words, 0)
        ids[:, 0] = bos
        ids[np.arange(n), length - 1] = eos
        flip = rng.random(n) < 0.4
        src = np.arange(n)
        src[flip] = rng.permutation(src[flip])                       # a noisy sample gets another noisy sample's caption
        data[name] = dict(pixels=px, ids=torch.from_numpy(ids[src]), label_id=None, clean=np.arange(n), noisy=src,
                          caption_tokens=length[src])
    n_tr = args.n_train
    if n_tr > args.db_limit:
        np.random.seed(0)                                             # run_lemon.py:81 seeds, :123 draws (first consumer of the stream)
        sel = np.random.choice(np.arange(n_tr), args.db_limit, replace=False)
        mask = np.zeros(n_tr, dtype=np.uint8)
        mask[sel] = 1
        data["train"]["db_index"], data["train"]["in_db"] = sel, mask
    return data


def bench_cifar(args, world, rank, dev):
    from lemon_amd import _lib
    from lemon_amd.clip import ClipConfig, LemonCLIP, encoder_flops
    from lemon_amd.pipeline import Embedder, FIXED_HPARAMS, GatherLog, run_hot_path

    cfg = ClipConfig.named(args.arch)
    model = LemonCLIP(cfg)                       # seeded random init (no checkpoints offline)
    # text micro-batch = a common divisor of the split sizes (5 000 at the headline shape): every micro-batch then has the
    # same row count, so the same prompt gets the same GEMM shape -> the same bits wherever it is embedded (a short tail
    # batch meets another hipBLASLt solution and differs in the last bit: 200 "distinct" prompt embeddings for 100 classes)
    import math
    tb = math.gcd(math.gcd(args.n_train, args.n_val), args.n_test)
    while tb > 8 * args.encoder_batch and tb % 2 == 0:
        tb //= 2
    text_batch = tb if min(2 * args.encoder_batch, 4000) <= tb <= 8 * args.encoder_batch else None
    coco = args.workload == "mscoco"
    if coco:
        assert world == 1, "the mscoco workload is a one-GPU measurement (the DB subset is drawn over the whole train split)"
        # captions run 8 ... 77 tokens: micro-batches of ~2 500 captions (up to 197 000 token rows at 77 tokens) sorted by length
        emb = Embedder(model, dev, batch_size=args.encoder_batch, text_batch_size=4 * args.encoder_batch,
                       length_bucketing=not args.no_length_bucketing)
        data = make_mscoco_like(args, cfg, rank, dev)
    else:
        emb = Embedder(model, dev, batch_size=args.encoder_batch, text_dedup=args.text_dedup, text_batch_size=text_batch)
        data = make_cifar_like(args, cfg, rank, dev)
    data["train"]["n_total"] = args.n_train * world
    algo = {"auto": None, "f32": _lib.ALGO_F32_MFMA, "bf16": _lib.ALGO_BF16_FILTER}[args.algo]
    n_scored = args.n_train + args.n_val + args.n_test

    prof = {"launches": 0, "kernel_ms": 0.0, "algo_flops": 0.0, "algo_bytes": 0.0}
    stage = {"embed_s": 0.0, "knn_score_s": 0.0}

    prof_txt = {"launches": 0, "kernel_ms": 0.0}

    def collect(db):
        # the dominant kernel = the image-side scan (50 000 distinct queries x the DB).  The text side folds its 50 000
        # queries to the C distinct class prompts first (query de-duplication, csrc/dedup.hip): a C-query launch that is
        # reported separately and not averaged into the roofline of the dominant launch
        p = db.index_img.profile_read()
        for k_ in prof:
            prof[k_] += p[k_]
        p = db.index_txt.profile_read()
        prof_txt["launches"] += p["launches"]; prof_txt["kernel_ms"] += p["kernel_ms"]

    glog = GatherLog() if (world > 1 or args.rccl_world1) else None   # HIP events around every all-gather of the timed region (read afterwards)

    def step(timers=None, events=False):
        return run_hot_path(emb, data, k=args.knn_k, dist_type=args.dist_type, hparams=FIXED_HPARAMS,
                            world_size=world, rank=rank, algo=algo, timers=timers, profile_index=events,
                            gather_log=glog if events else None)

    from lemon_amd import ops as _ops
    for _ in range(args.warmup):
        step()
    barrier_sync(world, dev)
    # HIP events bracket every launch of the step's dominant kernel (the hand-written split GEMM) and every scan-kernel launch
    # of the timed region: recorded on the launch stream inside the library, read back only after the region ends
    _ops.gemm_profiling(True)
    fb0 = (emb.fallback_batches, emb.fold_fallback_batches, emb.fallback_rows, emb.fold_fallback_rows)
    tok0 = emb.text_tokens_run
    t0 = time.perf_counter()
    timed = []
    gemm_prof = {"launches": 0, "kernel_ms": 0.0, "flops": 0.0}
    for _ in range(args.steps):
        recs, db = step(events=True)
        timed.append(db)
    barrier_sync(world, dev)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    fb1 = (emb.fallback_batches, emb.fold_fallback_batches, emb.fallback_rows, emb.fold_fallback_rows)
    text_tokens_per_caption = (emb.text_tokens_run - tok0) / max(args.steps * n_scored, 1)
    gp = _ops.gemm_profile_read()
    _ops.gemm_profiling(False)
    for k_ in gemm_prof:
        gemm_prof[k_] += gp[k_]
    for db_ in timed:
        collect(db_)
    del timed

    # one extra, untimed step with host syncs between stages: embed vs kNN+score wall split
    timers = {}
    recs, db = step(timers=timers)
    for k_ in stage:
        stage[k_] += timers[k_]
    info = db.index_img.last_search_info()
    # text tokens actually run: the longest prompt rounded up to the tower's bucket (clip.TextTower.seq_len_for)
    f_img, f_txt = encoder_flops(cfg, n_tokens_text=model.text.seq_len_for(int(data["train"]["ids"].argmax(-1).max().item())))
    if coco:
        # captions: FLOPs of the tokens the tower RAN (micro-batches sorted by length run their longest caption's bucket), averaged per
        # caption from a per-length table; the reference pads every caption to 77 tokens (run_lemon.py:151-154): `ref_padded` below
        f_txt_padded = encoder_flops(cfg, n_tokens_text=cfg.context_length)[1]
        Lbar = max(text_tokens_per_caption, 1.0)
        lo_, hi_ = int(np.floor(Lbar)), int(np.ceil(Lbar))
        f_lo, f_hi = encoder_flops(cfg, n_tokens_text=max(lo_, 1))[1], encoder_flops(cfg, n_tokens_text=max(hi_, 1))[1]
        f_txt = f_lo + (f_hi - f_lo) * (Lbar - lo_)

    value = n_scored * world * args.steps / elapsed
    from lemon_amd.ops import gemm_mode as _gm
    gemm_mode = _gm()
    gemm_notes = {
        "bf16x6": "the four GEMMs of every transformer block (QKV, output projection, fc1, fc2): lemon_linear_bf16x6 -- both fp32 operands "
                  "split EXACTLY into three bf16 parts, the six cross products of order <= 2 summed by one hipBLASLt bf16 GEMM with fp32 "
                  "accumulation (delivered error vs float64 at the fp32 GEMM's level, tests/test_gpu_parity.py); patch embedding / projections: lemon_linear_f32; "
                  "recorded solution per shape, bias / SiLU / residual epilogues; LEMON_GEMM = f32 | bf16x6 | f16x3 selects the mode",
        "f16x3": f"LEMON_MLP={_ops.mlp_mode()} (block = default: QKV, output projection, fc1, fc2 of every block; fused: fc1, fc2): hand-written split-fp16 GEMM "
                 "gemm_f16x3.hip (v_mfma_f32_16x16x32_f16, tile-major operands written by the preprocess kernel / attention / the GEMM epilogues, "
                 f"LDS-DMA ring; LEMON_LNFOLD={int(_ops.ln_fold_enabled())}: the two LayerNorms of a block folded into QKV / fc1, whose operands and row statistics "
                 "the output projection / fc2 epilogues write -- accuracy contract of the fold: a folded GEMM's error may exceed the fp32 GEMM's by the "
                 "factor sqrt(1 + (mean/sigma)^2) of the LayerNorm input row, at most 8.06 at |mean|/sigma = 8, beyond which the kernels poison the row and "
                 "the SAMPLE is embedded again with LayerNorm kernels (`fold_fallback_rows`); measured on real-checkpoint-like statistics -- gains of 30, "
                 "40-1000x outlier channels, rows up to |mean|/sigma 7.9 -- 0.7-1.8x the fp32 chain's error, tests/test_gpu_parity.py::"
                 "test_folded_layernorm_chain_on_real_checkpoint_statistics); "
                 f"LEMON_CHAIN_RES={int(_ops.chain_operand_residual())}: inside the chain the residual stream exists only as that operand (hi + lo 2^-11: 22 "
                 "significant bits per hop, read by the next GEMM as its operand and by the one behind it as its residual; fp32 is written by the last "
                 "chained block only); also the patch embedding and the last block's all-token QKV; the pooled rows of the "
                 "last block (and what the mode leaves out): lemon_linear_f16x3 on hipBLASLt -- both fp32 operands "
                 "split into two fp16 parts (hi = f16(v), lo = the exact remainder kept to 11 bits: 22 bits + sign), weights pre-scaled by a "
                 "power of two, hi.hi + hi.lo + lo.hi summed by one fp16 GEMM with fp32 accumulation (error vs float64 at the "
                 "fp32 GEMM's level, tests/test_gpu_parity.py); final projections: lemon_linear_f32; recorded solution per "
                 "shape, bias / SiLU / residual epilogues; LEMON_GEMM = f32 | bf16x6 | f16x3 selects the mode",
        "f32": "lemon_linear_f32 (hipBLASLt fp32, recorded solution per shape, SiLU/residual epilogues)"}
    gemm_note = gemm_notes[gemm_mode]
    line = {
        "metric": "label-error scores/sec (embed+kNN), CIFAR-100 noise=0.4",
        "value": value, "unit": "scores/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"f32": "f32", "f16x3": "f32-equivalent (fp16 x3 split GEMMs: hi.hi + hi.lo + lo.hi, fp32 accumulate, LayerNorms folded in: error <= fp32 GEMM x sqrt(1 + (mean/sigma)^2) <= 8.06, measured <= 1.8; everything else fp32)",
                  "bf16x6": "f32-equivalent (bf16 x6 split GEMMs, fp32 accumulate; everything else fp32)"}[gemm_mode],
        "data": "synthetic",
        "config": {
            "workload": f"CIFAR-100 shape per GPU: {args.n_train} train (=DB shard) + {args.n_val} val + {args.n_test} test "
                        f"image/prompt pairs ({'raw 32x32 uint8 images, bicubic resize to 224 + normalise in the step' if args.input == 'u8' else 'preprocessed float pixels'}), CLIP {args.arch} random-init fp32 encoder -> {cfg.embed_dim}-d, "
                        f"{args.dist_type} brute-force kNN k={args.knn_k} over the all-gathered {args.n_train * world}-row DB, "
                        "multimodal-neighbour scores (beta=gamma=5,tau1=0.1,tau2=5)",
            "noise": "pair-flip 0.4 (reference's 'asymmetric'; 'cat' is not defined for CIFAR upstream)",
            "encoder_batch": args.encoder_batch, "text_batch": emb.text_batch_size, "text_dedup": bool(args.text_dedup),
            "train_embedded_once": True, "parallelism": f"dp{world}+allgather",
            "gemm": gemm_note,
        },
        # samples of the TIMED region that were embedded a second time (pipeline.Embedder: fp16-range overflow -> bf16x6; a row beyond
        # the folded LayerNorm's mean bound -> LayerNorm kernels).  Non-zero = the step did part of its encoder work twice.
        "fallback_batches": fb1[0] - fb0[0], "fold_fallback_batches": fb1[1] - fb0[1],
        "fallback_rows": fb1[2] - fb0[2], "fold_fallback_rows": fb1[3] - fb0[3],
        "stages_s": {k_: v for k_, v in stage.items()},
        "encoder": {"bound": "mfma", "unit": "TFLOP/s", "gemm_mode": gemm_mode,
                    # fp32-equivalent peak of the mode: the 16-bit matrix peak divided by the products one fp32 product costs
                    "peak": {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "f16x3": PEAK_BF16_MFMA_TFLOPS / 3}[gemm_mode],
                    "achieved": (f_img + f_txt) * n_scored / max(stage["embed_s"], 1e-9) / 1e12,
                    "frac": (f_img + f_txt) * n_scored / max(stage["embed_s"], 1e-9) / 1e12
                            / {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "f16x3": PEAK_BF16_MFMA_TFLOPS / 3}[gemm_mode],
                    "note": "algorithmic forward FLOPs (img+txt) of the fp32 model / embed stage wall time (preprocess, LayerNorm, attention and split passes included), against the fp32-equivalent matrix peak of the GEMM mode; "
                            "in gemm_mode 'bf16x6' / 'f16x3' the four GEMMs of every block run as split GEMMs on the 16-bit matrix "
                            "cores (fp32-equivalent results; 16-bit MFMA peak / 6 = 416.7, / 3 = 833 TFLOP/s-equivalent), the patch "
                            "embedding with them; the two final projections (1 000 pooled rows per micro-batch) stay fp32 GEMMs"},
    }
    if coco:
        H_, W_ = (int(v) for v in args.image_hw.lower().split("x"))
        line["metric"] = "label-error scores/sec (embed+kNN), mscoco-shaped captions noise=0.4"
        line["scaling"] = "none (one GPU)"
        line["config"]["workload"] = (
            f"mscoco shape (BASELINE configs[2]): {args.n_train} train + {args.n_val} val + {args.n_test} test image/caption pairs; decoded "
            f"{H_}x{W_} uint8 images resident in HBM (bicubic resize to 224 + centre crop + normalise in the step), one DISTINCT caption per "
            f"image, 8 ... {cfg.context_length} tokens, CLIP {args.arch} random-init fp32 encoder -> {cfg.embed_dim}-d; DB = a random "
            f"{min(args.db_limit, args.n_train)}-row subset of the train split (run_lemon.py:121-127), {args.dist_type} brute-force kNN "
            f"k={args.knn_k}, every one of the {n_scored} samples scored (train queries with self-exclusion where they are DB rows)")
        line["config"]["noise"] = "40 % of the samples carry another sample's caption (synthetic stand-in for 'cat' noise, lib/datasets/utils.py:300-309)"
        line["config"]["text_dedup"] = False
        line["config"]["length_bucketing"] = bool(emb.length_bucketing)
        line["config"]["text_tokens_run_per_caption"] = text_tokens_per_caption
        line["config"]["text_tokens_mean_caption"] = float(np.mean(np.concatenate([data[n_]["caption_tokens"] for n_ in data])))
        line["encoder"]["text_flops_note"] = ("text FLOPs = the tokens the tower ran (captions sorted by length, a micro-batch runs its longest caption's "
                                              "8-token bucket); padded to 77 tokens as upstream the same captions cost "
                                              f"{f_txt_padded / 1e9:.2f} GFLOP each instead of {f_txt / 1e9:.2f}")
        # the hand-written GEMM per tower (the same kernels meet other shapes in the text tower: width 512, 3 ... 20 x fewer rows per
        # launch): one untimed image-only and one text-only pass over (at most 20 000 samples of) the train split, HIP events around
        # every launch
        nbt = min(20000, args.n_train)
        by_tower = {}
        for tname, fn in (("image", lambda: emb.embed_images(data["train"]["pixels"][:nbt])), ("text", lambda: emb.embed_texts(data["train"]["ids"][:nbt]))):
            fn(); torch.cuda.synchronize(dev)
            _ops.gemm_profiling(True)
            fn()
            gpt = _ops.gemm_profile_read()
            _ops.gemm_profiling(False)
            if gpt["launches"]:
                tf_ = gpt["flops"] / (gpt["kernel_ms"] / 1e3) / 1e12
                by_tower[tname] = {"launches": gpt["launches"], "avg_launch_ms": gpt["kernel_ms"] / gpt["launches"], "achieved": tf_,
                                   "frac": tf_ / PEAK_BF16_MFMA_TFLOPS}
        line["gemm_by_tower"] = by_tower
    if world == 1 and not args.text_dedup and not args.no_f32_gemm_check and not coco:
        # what `python -m lemon_amd.run_lemon` does by default (cli_common: --no_text_dedup turns it off): every DISTINCT prompt is
        # embedded once and gathered -- 100 prompts instead of 50 000 on this workload.  One extra step, an untimed region of its
        # own; NOT the headline (which encodes every sample's prompt, as upstream does, run_lemon.py:140-161,207-233)
        emb_d = Embedder(model, dev, batch_size=args.encoder_batch, text_dedup=True, text_batch_size=text_batch)
        td = {}
        run_hot_path(emb_d, data, k=args.knn_k, dist_type=args.dist_type, hparams=FIXED_HPARAMS, world_size=world, rank=rank, algo=algo)      # warm
        recs_d, _ = run_hot_path(emb_d, data, k=args.knn_k, dist_type=args.dist_type, hparams=FIXED_HPARAMS, world_size=world, rank=rank,
                                 algo=algo, timers=td)
        line["product_default_text_dedup"] = {
            "scores_per_s": n_scored / max(td["embed_s"] + td["knn_score_s"], 1e-9), "embed_s": td["embed_s"],
            "max_abs_val_score_diff_vs_headline": float((recs_d["val"]["score"] - recs["val"]["score"]).abs().max().item()),
            "note": "run_lemon's default text path (distinct prompts embedded once); reported beside the headline, never as `value`"}
        del recs_d, emb_d
    if gemm_mode != "f32" and world == 1 and not args.no_f32_gemm_check and not coco:
        # the same step once more, untimed region of its own, with every GEMM on the fp32 matrix cores (LEMON_GEMM=f32): what
        # the split GEMMs buy, and the score difference between the two modes on the val split
        os.environ["LEMON_GEMM"] = "f32"
        t32 = {}
        eb_ = emb.batch_size
        emb.batch_size = min(eb_, 2620)          # (the fp32 library GEMMs' recorded solutions are keyed on 2 620-image micro-batches)
        recs32, _ = step(timers=t32)
        emb.batch_size = eb_
        os.environ["LEMON_GEMM"] = gemm_mode
        ds_ = (recs32["val"]["score"] - recs["val"]["score"]).abs().max().item()
        line["value_f32_gemm_mode"] = n_scored / max(t32["embed_s"] + t32["knn_score_s"], 1e-9)     # scores/s with every GEMM on the fp32 matrix cores
        line["encoder_f32_gemm_mode"] = {"embed_s": t32["embed_s"], "achieved": (f_img + f_txt) * n_scored / max(t32["embed_s"], 1e-9) / 1e12,
                                         "unit": "TFLOP/s", "max_abs_val_score_diff_vs_split": ds_,
                                         "max_abs_val_embedding_diff_vs_split": float((recs32["val"]["emb_img"] - recs["val"]["emb_img"]).abs().max().item())}
        del recs32
    # ---- roofline: the step's DOMINANT kernel.  With the default GEMM mode that is the hand-written split-fp16 GEMM
    # (k_gemm_f16x3t16: fc1 / fc2 and, LEMON_MLP=block, QKV / output projection of every block); the kNN scan -- the path's
    # own kernel, ~1 % of this step -- is reported under "knn".
    if gemm_prof["launches"]:
        gsec = gemm_prof["kernel_ms"] / 1e3
        gtf = gemm_prof["flops"] / gsec / 1e12
        gp_, gsrc = pmc_gemm_summary()
        line["roofline"] = {
            "kernel": "k_gemm_f16x3t16 (lemon_linear_f16x3t: v_mfma_f32_16x16x32_f16, 128 x 256 tiles)",
            "bound": "mfma", "achieved": gtf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": gtf / PEAK_BF16_MFMA_TFLOPS,
            "flops_note": "the kernel's executed fp16 arithmetic, 2 m n 3k per launch (three fp16 products per fp32 product), summed over the "
                          "launches of the timed region / their summed HIP-event durations on the launch stream",
            "launches": gemm_prof["launches"], "avg_launch_ms": gemm_prof["kernel_ms"] / gemm_prof["launches"],
            "share_of_timed_region": gsec / max(elapsed, 1e-9),
            "traffic": gp_.get("traffic"), "traffic_source": gsrc if gp_.get("traffic") is not None else None,
            "mfma_busy_pmc": gp_.get("mfma_busy"), "clock_GHz_pmc": gp_.get("clock_GHz"), "pmc_source": gsrc,
            "pmc_note": "counter figures (FETCH_SIZE x 2 + WRITE_SIZE per launch; SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024); "
                        "GRBM_GUI_ACTIVE / 8 / duration) of this kernel's longest-running launch group in the COMMITTED rocprofv3 passes of this "
                        "command, not of this run",
        }
    if prof["launches"]:
        sec = prof["kernel_ms"] / 1e3
        tf = prof["algo_flops"] / sec / 1e12
        traffic, traffic_src = pmc_traffic("k_scan_f32")
        pm_, _ = pmc_summary("k_scan_f32")
        f32scan = info["algo"] == _lib.ALGO_F32_MFMA
        hbm_model = prof["algo_bytes"] / sec / 1e9
        knn = {
            "kernel": "k_scan_f32" if f32scan else "k_scan_bf16",
            "bound": "mfma", "achieved": tf,
            "peak": PEAK_F32_MFMA_TFLOPS if f32scan else PEAK_BF16_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / (PEAK_F32_MFMA_TFLOPS if f32scan else PEAK_BF16_MFMA_TFLOPS),
            "traffic": traffic, "traffic_source": traffic_src, "launches": prof["launches"],
            "mfma_busy_pmc": (float(pm_["SQ_VALU_MFMA_BUSY_CYCLES"]["value_KB"]) / (float(pm_["GRBM_GUI_ACTIVE"]["value_KB"]) / 8.0 * 1024.0)
                              if "SQ_VALU_MFMA_BUSY_CYCLES" in pm_ and "GRBM_GUI_ACTIVE" in pm_ and f32scan else None),
            "avg_launch_ms": prof["kernel_ms"] / prof["launches"],
            "share_of_timed_region": sec / max(elapsed, 1e-9),
            "text_side": {"distinct_queries": db.index_txt.last_search_info()["nq_distinct"], "launches": prof_txt["launches"],
                          "avg_launch_ms": prof_txt["kernel_ms"] / max(prof_txt["launches"], 1)},
            "hbm_scan_model": hbm_model_object(info["query_panel"], hbm_model),
        }
        line["knn"] = knn
        if "roofline" not in line:            # GEMM modes without the hand-written kernel: the scan is what this file can time
            line["roofline"] = knn
    if world > 1 or args.rccl_world1:
        line["exchange"] = exchange_report(glog, world, info, db, args.steps)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if coco:
            cores = __import__("oracle.oracle", fromlist=["usable_cores"]).usable_cores()
            line["cpu_baseline"] = cpu_reference_style(args, cfg, data, recs, db, n_scored, cores, 0.0)
            line["cpu_baseline"]["preprocess_note"] = "image decoding / resizing not timed on the CPU side (generous to the CPU)"
        else:
            line["cpu_baseline"], line["auroc_check"] = cpu_baseline_cifar(args, cfg, model, data, recs, db, n_scored)
    if world == 1 and not coco and not (args.no_knn_1m and args.no_mscoco):
        del recs, db, data, emb, model
        torch.cuda.empty_cache()
    if world == 1 and not args.no_mscoco and not coco:
        # BASELINE configs[2] (caption-shaped: every text row distinct, 8 ... 77 tokens, ViT-B/16, DB = random subset) at a BOUNDED
        # size inside the same driver run; the full-size line is committed under profiles/ (tools/gpu_profile.sh)
        ma = argparse.Namespace(**{**vars(args), "workload": "mscoco", "arch": "vit-b-16", "n_train": 8000, "n_val": 500, "n_test": 500,
                                   "db_limit": 5000, "encoder_batch": 664, "steps": 2, "warmup": 1, "no_cpu_baseline": True,
                                   "no_knn_1m": True, "no_mscoco": True, "no_f32_gemm_check": True, "text_dedup": False, "algo": "auto",
                                   "input": "u8", "knn_k": 50, "dist_type": "cosine", "rccl_world1": False})
        m1 = bench_cifar(ma, world, rank, dev)
        line["mscoco"] = {k_: m1[k_] for k_ in ("metric", "value", "unit", "ms_per_step", "config", "stages_s", "encoder", "roofline",
                                                  "gemm_by_tower", "fallback_rows", "fold_fallback_rows") if k_ in m1}
        torch.cuda.empty_cache()
    if world == 1 and not args.no_knn_1m and not coco:
        # north_star's second target inside the same driver-timed run: the 1M x 768, k=50 self-join on this GPU
        ka = argparse.Namespace(**{**vars(args), "knn_n": 1_000_000, "knn_d": 768, "knn_k": 50, "steps": 1, "warmup": 0,
                                   "algo": "auto", "dist_type": "cosine"})
        k1 = bench_knn(ka, world, rank, dev)
        line["knn_1m"] = {"workload": k1["config"]["workload"], "ms": k1["ms_per_step"], "scores_per_s": k1["value"],
                          "roofline": k1["roofline"], "cpu_baseline": k1.get("cpu_baseline")}
    return line


def exchange_report(glog, world, info, db, steps):
    """N > 1: what the DB all-gather cost on rank 0 (per step, per array: ms, algorithm and bus GB/s) and which scan every rank
    ran against the gathered DB -- so that a scaling record explains itself."""
    import torch.distributed as dist
    from lemon_amd import _lib
    names = {_lib.ALGO_F32_MFMA: "f32", _lib.ALGO_BF16_FILTER: "bf16"}
    mine = {"img": names.get(info["algo"], str(info["algo"])),
            "txt": names.get(db.index_txt.last_search_info()["algo"], "?"),
            "txt_distinct_queries": int(db.index_txt.last_search_info()["nq_distinct"]), "db_rows": int(db.index_img.ntotal)}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    rep = glog.summary() if glog is not None else {"allgather_ms": 0.0, "arrays": []}
    rep["allgather_ms_per_step"] = rep["allgather_ms"] / max(steps, 1)
    rep["backend"] = dist.get_backend()
    rep["knn_algo_per_rank"] = per_rank
    rep["note"] = "HIP events around all_gather_into_tensor on rank 0; busbw = algbw x (W-1)/W (nccl-tests convention)"
    return rep


def cpu_baseline_cifar(args, cfg, model, data, recs, db, n_scored):
    """The CPU port (oracle/ + the same CLIP module on CPU fp32) timed on this host's cores on a bounded
    sample of the same workload, scaled to scores/s.  Also the AUROC parity check GPU vs oracle on the
    val split (same embeddings)."""
    from oracle import oracle as o
    from lemon_amd.pipeline import FIXED_HPARAMS
    cores = o.usable_cores()             # affinity mask / cgroup quota, not os.cpu_count()
    torch.set_num_threads(cores)
    o.set_threads(cores)
    cpu_model = type(model)(cfg)
    cpu_model.load_state_dict(model.state_dict())
    cpu_model = cpu_model.float().eval()
    ni = min(args.cpu_sample_images, args.n_val)
    px = data["val"]["pixels"][:ni].cpu()
    if px.dtype == torch.uint8:      # the reference's CPU path: PIL bicubic resize + normalise per image
        from PIL import Image
        from lemon_amd.data import generic_transform
        t_pre = time.perf_counter()
        px = torch.stack([generic_transform(Image.fromarray(im.numpy()), cfg.image_size) for im in px])
        t_pre = (time.perf_counter() - t_pre) / ni
    else:
        t_pre = 0.0
    ids = data["val"]["ids"][:ni].cpu()
    cpu_model.encode_image(px[:8])                              # warm
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(0, ni, 32):
            cpu_model.encode_image(px[i:i + 32]); cpu_model.encode_text(ids[i:i + 32])
    t_embed = (time.perf_counter() - t0) / ni                   # s per (image, prompt) pair
    # kNN + per-sample quantities + scores: the oracle on the first nqs val queries against the full DB
    nqs = min(args.cpu_sample_queries, args.n_val)
    img_tr, txt_tr = db.img.cpu().numpy(), db.txt.cpu().numpy()
    rv = recs["val"]
    q_img = rv["emb_img"][:nqs].cpu().numpy()      # the very embeddings the GPU scores came from
    q_txt = rv["emb_txt"][:nqs].cpu().numpy()
    t0 = time.perf_counter()
    ref = o.neighbors(args.dist_type, img_tr, txt_tr, q_img, q_txt, args.knn_k)
    sref = o.score(ref, FIXED_HPARAMS)
    t_knn = (time.perf_counter() - t0) / nqs                    # s per query (DB fixed at n_train rows)
    t_pre /= cores                                               # PIL work spread over all cores (generous to the CPU)
    per_sample = t_pre + t_embed + t_knn
    y = (data["val"]["clean"] != data["val"]["noisy"])[:nqs]
    sgpu = rv["score"][:nqs].cpu().numpy()
    same_sets = bool(np.array_equal(rv["I_n"][:nqs].cpu().numpy(), ref["I_n"]) and
                     np.array_equal(rv["I_m"][:nqs].cpu().numpy(), ref["I_m"]))
    arrays_equal = all(np.array_equal(rv[key][:nqs].cpu().numpy(), ref[key]) for key in
                       ("d_1", "D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m"))
    auroc = {"gpu": o.auroc(y, sgpu), "oracle": o.auroc(y, sref), "n": int(nqs),
             "max_abs_score_diff": float(np.abs(sgpu - sref).max()), "topk_sets_identical": same_sets,
             "per_sample_arrays_bit_exact": bool(arrays_equal)}
    # The towers are random-init: image and text embeddings are not aligned, so d_1 is noise and the fixed-hparam score
    # (d_1 + 5 d_n + 5 d_m) sits near chance.  What CAN carry signal is the neighbour structure (same-class images are
    # planted near each other, the neighbours' prompts vote): the same parity check on score variants that weight it
    # (SURVEY 8d metrics (i)-(iii)) -- GPU (lemon_score on the GPU records) against the oracle, each to 3 decimals.
    from lemon_amd import ops
    rec_gpu = {key: rv[key][:nqs] for key in ("d_1", "D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m")}
    zero_d1 = dict(rec_gpu, d_1=torch.zeros_like(rec_gpu["d_1"]))
    ref_zero = dict(ref, d_1=np.zeros_like(ref["d_1"]))
    variants = {
        "d1_only (--ablation multimodal_baseline)": (dict(beta=0, gamma=0, tau_1_n=0, tau_2_n=0, tau_1_m=0, tau_2_m=0), rec_gpu, ref),
        "beta=gamma=100": (dict(FIXED_HPARAMS, beta=100.0, gamma=100.0), rec_gpu, ref),
        "neighbour terms only (--ablation d1, beta=gamma=1, tau=0)": (dict(beta=1, gamma=1, tau_1_n=0, tau_2_n=0, tau_1_m=0, tau_2_m=0), zero_d1, ref_zero),
        "image-neighbour term only (d1 zeroed, beta=1)": (dict(beta=1, gamma=0, tau_1_n=0, tau_2_n=0, tau_1_m=0, tau_2_m=0), zero_d1, ref_zero),
    }
    auroc["variants"] = {}
    for name, (hp, rg, rr) in variants.items():
        sg = ops.lemon_score(rg, hp).cpu().numpy()
        so = o.score(rr, hp)
        ag, ao = o.auroc(y, sg), o.auroc(y, so)
        auroc["variants"][name] = {"gpu": ag, "oracle": ao, "equal_to_3_decimals": bool(round(ag, 3) == round(ao, 3)),
                                   "max_abs_score_diff": float(np.abs(sg - so).max())}
    auroc["informative"] = bool(max(v["gpu"] for v in auroc["variants"].values()) > 0.6)
    port = {
        "value": 1.0 / per_sample, "unit": "scores/s", "cores": cores, "kind": "port",
        "sample": f"{ni} images through PIL generic_transform (time / {cores} cores) and {ni} image+prompt pairs through the same CLIP module on CPU fp32 (torch, {cores} threads) "
                  f"+ oracle kNN/neighbours/score for {nqs} val queries against the full {args.n_train}-row DB "
                  f"(OpenMP, {cores} threads); per-sample times added and inverted; train embedded once as on the GPU",
        "preprocess_s_per_sample": t_pre, "embed_s_per_sample": t_embed, "knn_score_s_per_sample": t_knn,
    }
    base = cpu_reference_style(args, cfg, data, recs, db, n_scored, cores, t_pre)
    base["port_oracle"] = port
    return base, auroc


def cpu_reference_style(args, cfg, data, recs, db, n_scored, cores, t_pre):
    """The CPU baseline SURVEY 8d / BASELINE.md section 3 define, leg by leg, on this host's cores (oracle/reference_loop.py):
      encoder   HF transformers CLIPModel (the class lib/models/downstream_models.py:30-41 wraps) of the same architecture,
                fp32 on CPU, batch 128, 512 image + prompt pairs timed (random-init weights: same FLOPs);
      kNN       float32 torch.mm of a 128-query batch against the full DB + exact top-k (the faiss IndexFlat stand-in,
                run_lemon.py:235-236), both modalities;
      scoring   (a) the faithful per-sample Python loop (run_lemon.py:238-307, O(N) membership test included), (b) numpy vectorised.
    `value` = reference-equivalent end to end: every scored sample preprocessed and embedded, the train split embedded TWICE
    (run_lemon.py:137-161 and :198-233), searched in 128-query batches and scored by the loop; extrapolated per sample."""
    import torch as _t
    from oracle import reference_loop as rl
    _t.set_num_threads(cores)
    legs = {}
    # ---- encoder: HF CLIPModel on CPU ----
    n_enc, bs = 512, 128
    try:
        from transformers import CLIPConfig, CLIPModel
        hcfg = CLIPConfig(projection_dim=cfg.embed_dim,
                          vision_config=dict(hidden_size=cfg.vision.width, num_hidden_layers=cfg.vision.layers, num_attention_heads=cfg.vision.heads,
                                             intermediate_size=cfg.vision.mlp, image_size=cfg.image_size, patch_size=cfg.patch_size),
                          text_config=dict(hidden_size=cfg.text.width, num_hidden_layers=cfg.text.layers, num_attention_heads=cfg.text.heads,
                                           intermediate_size=cfg.text.mlp, vocab_size=cfg.vocab_size, max_position_embeddings=cfg.context_length,
                                           eos_token_id=2, bos_token_id=0, pad_token_id=1))
        _t.manual_seed(0)
        hf = CLIPModel(hcfg).eval()
        px = _t.randn(n_enc, 3, cfg.image_size, cfg.image_size)
        ids = data["val"]["ids"][:n_enc].cpu().long()
        if ids.shape[0] < n_enc:
            ids = ids.repeat((n_enc + ids.shape[0] - 1) // ids.shape[0], 1)[:n_enc]
        mask = (_t.arange(ids.shape[1])[None, :] <= ids.argmax(-1)[:, None]).long()
        with _t.no_grad():
            hf.get_image_features(pixel_values=px[:8]); hf.get_text_features(input_ids=ids[:8], attention_mask=mask[:8])     # warm
            t0 = time.perf_counter()
            for i in range(0, n_enc, bs):
                hf.get_image_features(pixel_values=px[i:i + bs])
            t_img = (time.perf_counter() - t0) / n_enc
            t0 = time.perf_counter()
            for i in range(0, n_enc, bs):
                hf.get_text_features(input_ids=ids[i:i + bs], attention_mask=mask[i:i + bs])
            t_txt = (time.perf_counter() - t0) / n_enc
        del hf
        legs["encoder"] = {"kind": "HF transformers CLIPModel, fp32 on CPU, batch 128 (lib/models/downstream_models.py:30-41)", "samples": n_enc,
                           "image_s_per_sample": t_img, "text_s_per_sample": t_txt, "pairs_per_s": 1.0 / (t_img + t_txt)}
        t_embed = t_img + t_txt
    except Exception as e:       # transformers missing / incompatible: say so, fall back to the module timed in the port
        legs["encoder"] = {"kind": "unavailable", "error": repr(e)[:200]}
        t_embed = None
    # ---- kNN + scoring on the very embeddings the GPU scored ----
    img_tr, txt_tr = db.img.cpu(), db.txt.cpu()
    dists_tr = (1 - (txt_tr * img_tr).sum(1)) if args.dist_type == "cosine" else ((txt_tr - img_tr) ** 2).sum(1)
    ii, it = rl.FlatIndexTorch(img_tr.shape[1], args.dist_type), rl.FlatIndexTorch(img_tr.shape[1], args.dist_type)
    t0 = time.perf_counter()
    ii.add(img_tr.numpy()); it.add(txt_tr.numpy())
    t_add = time.perf_counter() - t0
    n_tr_q = min(2048, recs["train"]["emb_img"].shape[0])
    n_va_q = min(1024, recs["val"]["emb_img"].shape[0])
    splits = (("train", recs["train"], n_tr_q), ("val", recs["val"], n_va_q))
    # single GPU: the DB is the whole train split in order, or (mscoco) the drawn subset of it
    in_compr = np.asarray(data["train"].get("db_index")) if data["train"].get("db_index") is not None else np.arange(img_tr.shape[0])
    pos_in_db = np.full(recs["train"]["emb_img"].shape[0], -1, dtype=np.int64)
    pos_in_db[in_compr] = np.arange(len(in_compr))
    t_search = t_loop = t_vec = 0.0
    nq_tot = 0
    rows_same = 0
    d1_diff = 0.0
    adj = {"rows": 0, "rows_differing": 0, "gpu_ok": 0.0, "cpu_ok": 0.0, "max_gap_at_swap": 0.0}
    for sname, rv, nq in splits:
        qi, qt = rv["emb_img"][:nq].cpu(), rv["emb_txt"][:nq].cpu()
        kk = args.knn_k + (sname == "train")
        t0 = time.perf_counter()
        for lo in range(0, nq, 128):
            ii.search(qi[lo:lo + 128].numpy(), kk); it.search(qt[lo:lo + 128].numpy(), kk)
        t_search += time.perf_counter() - t0
        t0 = time.perf_counter()
        logs = rl.per_sample_loop(sname, qi, qt, img_tr, txt_tr, dists_tr, ii, it, args.knn_k, 128, in_compr, args.dist_type)
        t_loop += time.perf_counter() - t0
        t0 = time.perf_counter()
        vec = rl.vectorised(sname, qi.numpy(), qt.numpy(), img_tr.numpy(), txt_tr.numpy(), dists_tr.numpy(), ii, it, args.knn_k, 128,
                            pos_in_db[:nq] >= 0, args.dist_type)
        t_vec += time.perf_counter() - t0
        nq_tot += nq
        # against the GPU records: image-side neighbour sets and d_1.  (torch.mm's float32 summation order is not the chain order
        # of the exact scan: a near-tie at the k-th place can fall the other way, so the share of identical rows is reported)
        I_gpu, I_cpu = rv["I_n"][:nq].cpu().numpy(), rl.stack(logs, "I_n")
        rows_same += int((np.sort(I_cpu, 1) == np.sort(I_gpu, 1)).all(1).sum())
        d1_diff = max(d1_diff, float(np.abs(rl.stack(logs, "d_1") - rv["d_1"][:nq].cpu().numpy()).max()))
        # which side is right where they differ: float64 scores over the whole DB for every differing query (train: self excluded)
        a_ = rl.adjudicate_near_ties(qi.numpy(), img_tr.numpy(), I_gpu, I_cpu, args.dist_type,
                                     exclude=(pos_in_db[:nq] if sname == "train" else None))
        adj["rows"] += a_["rows"]; adj["rows_differing"] += a_["rows_differing"]
        adj["gpu_ok"] += a_["rows_a_equals_f64_set"] * a_["rows"]; adj["cpu_ok"] += a_["rows_b_equals_f64_set"] * a_["rows"]
        adj["max_gap_at_swap"] = max(adj["max_gap_at_swap"], a_["max_gap_at_swap"])
    t_loop_only = max(t_loop - t_search, 0.0)
    legs["knn"] = {"kind": "float32 torch.mm + exact top-k per 128-query batch, both modalities (faiss IndexFlat stand-in, run_lemon.py:235-236)",
                   "queries": nq_tot, "db_rows": int(img_tr.shape[0]), "s_per_query": t_search / nq_tot, "queries_per_s": nq_tot / t_search,
                   "index_add_s": t_add}
    legs["scoring_python_loop"] = {"kind": "faithful per-sample Python loop incl. the O(N) membership test (run_lemon.py:238-307); search time subtracted",
                                   "samples": nq_tot, "s_per_sample": t_loop_only / nq_tot}
    legs["scoring_vectorised"] = {"kind": "numpy vectorised twin of the loop, searches included", "samples": nq_tot,
                                  "s_per_sample": t_vec / nq_tot}
    embeds_per_scored = (2 * args.n_train + args.n_val + args.n_test) / n_scored          # the reference embeds train twice
    if t_embed is None:
        return {"value": None, "unit": "scores/s", "cores": cores, "kind": "port", "legs": legs, "sample": "HF CLIPModel unavailable"}
    per_ref = (t_pre + t_embed) * embeds_per_scored + t_search / nq_tot + t_loop_only / nq_tot
    per_vec = (t_pre + t_embed) * embeds_per_scored + t_vec / nq_tot
    return {
        # ("port": oracle/ holds no compiled or imported reference code -- the reference cannot run here, SURVEY 8c --, but this IS the
        # reference-style leg SURVEY 8d defines: HF CLIPModel + torch.mm / top-k in 128-query batches + the per-sample Python loop)
        "value": 1.0 / per_ref, "unit": "scores/s", "cores": cores, "kind": "port", "style": "reference-style restatement (SURVEY 8d recipe)",
        "sample": f"reference-style CPU path restated (oracle/reference_loop.py), per-sample times of bounded samples added and inverted: PIL "
                  f"generic_transform (time / {cores} cores) + HF CLIPModel fp32 batch 128 on {n_enc} image+prompt pairs, both x {embeds_per_scored:.1f} "
                  f"(train embedded twice, as upstream) + torch.mm/top-k search of {nq_tot} queries (128 per call) against the {int(img_tr.shape[0])}-row DB "
                  f"+ the per-sample Python loop on those queries; torch {cores} threads",
        "value_with_vectorised_scoring": 1.0 / per_vec,
        "embeds_per_scored_sample": embeds_per_scored, "preprocess_s_per_sample": t_pre, "legs": legs,
        "agreement_with_gpu_on_sample": {
            "rows_with_identical_image_neighbour_sets": rows_same / nq_tot, "max_abs_d1_diff": d1_diff,
            # float64 adjudication of every differing row (oracle/reference_loop.adjudicate_near_ties): the share of rows whose
            # set IS the float64-exact top-k, per side, and the largest float64 score distance between rows in dispute
            "rows_differing": adj["rows_differing"], "rows_gpu_equals_f64_set": adj["gpu_ok"] / max(adj["rows"], 1),
            "rows_cpu_equals_f64_set": adj["cpu_ok"] / max(adj["rows"], 1), "max_gap_at_swap": adj["max_gap_at_swap"],
            "note": "two float32 searches with different summation orders (GPU: fmaf chain; torch.mm: blocked) swap neighbours only where "
                    "the k-th and (k+1)-th float64 scores are closer than float32 noise: max_gap_at_swap is that distance"},
    }


def bench_knn(args, world, rank, dev):
    """configs[3] shape: synthetic N x d unit embeddings per modality, self-join, k+1 with self-exclusion.
    Strong-scaled over ranks by query sharding (DB replicated after an all-gather of the shards)."""
    from lemon_amd import _lib, ops
    from lemon_amd.neighbors import LemonDB
    from lemon_amd.pipeline import FIXED_HPARAMS, GatherLog, all_gather_rows, shard_bounds
    n, d, k = args.knn_n, args.knn_d, args.knn_k
    lo, hi = shard_bounds(n, world, rank)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    img = ops.normalize_vectors(torch.randn((hi - lo, d), generator=g, device=dev))
    txt = ops.normalize_vectors(torch.randn((hi - lo, d), generator=g, device=dev))
    algo = {"auto": None, "f32": _lib.ALGO_F32_MFMA, "bf16": _lib.ALGO_BF16_FILTER}[args.algo]
    prof = {"launches": 0, "kernel_ms": 0.0, "algo_flops": 0.0, "algo_bytes": 0.0}

    glog = GatherLog() if (world > 1 or args.rccl_world1) else None

    def step(events):
        lg = glog if events else None
        db = LemonDB(all_gather_rows(img, n, log=lg, name="emb_img"), all_gather_rows(txt, n, log=lg, name="emb_txt"),
                     args.dist_type, algo=algo)
        if events:
            db.index_img.set_profiling(True); db.index_txt.set_profiling(True)
        rec = db.neighbors(img, txt, k, drop_self=True, return_indices=False)
        s = ops.lemon_score(rec, FIXED_HPARAMS)
        return db, s

    for _ in range(args.warmup):
        step(False)
    barrier_sync(world, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        db, s = step(True)       # events recorded on the launch stream, read after the region
        torch.cuda.synchronize(dev)
        for ix in (db.index_img, db.index_txt):
            p = ix.profile_read()
            for k_ in prof:
                prof[k_] += p[k_]
    barrier_sync(world, dev)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    info = db.index_img.last_search_info()
    sec = prof["kernel_ms"] / 1e3
    f32 = info["algo"] == _lib.ALGO_F32_MFMA
    peak = PEAK_F32_MFMA_TFLOPS if f32 else PEAK_BF16_MFMA_TFLOPS
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_knn(args, db, img, txt, s, k)
    line = {
        "metric": "label-error scores/sec (kNN+score only), synthetic embeddings", "value": n * args.steps / elapsed,
        "unit": "scores/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic {n}x{d} unit embeddings per modality, self-join k={k} (+1 self-exclusion), "
                               f"{args.dist_type}, query-sharded over {world} GPU(s), DB all-gathered"},
        "roofline": {"kernel": "k_scan_f32" if f32 else "k_scan_f16_qs4 + k_bf16_final (fp16 filter scan on v_mfma_f32_16x16x32_f16 + exact fp32 re-score)", "bound": "mfma",
                     "achieved": prof["algo_flops"] / sec / 1e12, "peak": peak, "unit": "TFLOP/s",
                     "frac": prof["algo_flops"] / sec / 1e12 / peak, "traffic": None,
                     "launches": prof["launches"], "avg_launch_ms": prof["kernel_ms"] / max(prof["launches"], 1),
                     "hbm_scan_model": hbm_scan_model_object(info["query_panel"], prof["algo_bytes"] / sec / 1e9)},
    }
    if not f32 and n == 1000000 and d == 768:
        # counter figures of the steady-state chunk launch from the committed PMC passes of THIS command (tools/gpu_profile.sh):
        # not a measurement of the current run, the source file is named
        pm, src = pmc_summary("k_scan_f16_qs4", "knn_1000000x768_pmc.csv")
        if not pm:                               # (no round-5 summary committed yet: the round-4 one describes k_scan_bf16_qs2)
            pm, src = pmc_summary("k_scan_bf16_qs2", "knn_1000000x768_pmc.csv")
        if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
            line["roofline"]["traffic"] = (2.0 * float(pm["FETCH_SIZE"]["value_KB"]) + float(pm["WRITE_SIZE"]["value_KB"])) * 1024.0
            line["roofline"]["traffic_source"] = src
            line["roofline"]["traffic_note"] = ("per steady-state chunk launch (FETCH_SIZE x 2 + WRITE_SIZE; fabric reads served by the "
                                                "Infinity Cache / L2 are counted): compare with avg_launch_ms, not with the whole job")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in pm and "GRBM_GUI_ACTIVE" in pm:
            line["roofline"]["mfma_busy_pmc"] = float(pm["SQ_VALU_MFMA_BUSY_CYCLES"]["value_KB"]) / (float(pm["GRBM_GUI_ACTIVE"]["value_KB"]) / 8.0 * 1024.0)
    if not f32:
        sus, src = sustained_bf16_peak()
        if sus:
            line["roofline"]["peak_sustained_random_operands"] = {"value": sus, "unit": "TFLOP/s", "source": src,
                                                                  "frac": prof["algo_flops"] / sec / 1e12 / sus}
    if world > 1 or args.rccl_world1:
        line["exchange"] = exchange_report(glog, world, info, db, args.steps)
    if cpu is not None:
        line["cpu_baseline"] = cpu
    return line


def cpu_baseline_knn(args, db, img, txt, s_gpu, k):
    """The CPU oracle (AVX2 + OpenMP C restatement, `kind: port`) on a bounded query slice of the SAME self-join: the first
    nq rows as queries against the full DB, both modalities, neighbours + scores; also checks the GPU result on the slice."""
    from oracle import oracle as o
    from lemon_amd.pipeline import FIXED_HPARAMS
    cores = o.usable_cores()
    o.set_threads(cores)
    nq = min(args.knn_cpu_queries, img.shape[0])
    img_tr, txt_tr = db.img.cpu().numpy(), db.txt.cpu().numpy()
    t0 = time.perf_counter()
    ref = o.neighbors(args.dist_type, img_tr, txt_tr, img_tr[:nq], txt_tr[:nq], k, drop_self=True)
    sref = o.score(ref, FIXED_HPARAMS)
    dt = time.perf_counter() - t0
    return {"value": nq / dt, "unit": "scores/s", "cores": cores, "kind": "port",
            "sample": f"oracle neighbours+score for the first {nq} of {img.shape[0]} queries against the full {img_tr.shape[0]}x{img_tr.shape[1]} "
                      f"DB, both modalities, k={k}+1 ({dt:.1f} s, OpenMP {cores} threads); scaled per query",
            "gpu_scores_match_on_slice": bool(np.abs(s_gpu[:nq].cpu().numpy() - sref).max() <= 1e-9)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)               # before any GPU call in this process
    if args.rccl_world1 and args.gpus != 1:
        raise SystemExit("bench.py: --rccl_world1 is the one-GPU rehearsal of the exchange step (use it with --gpus 1)")
    world, rank, dev = init_dist(args)
    from lemon_amd import _lib
    _lib.load()                                   # fail loudly if the HIP library is missing
    line = bench_knn(args, world, rank, dev) if args.workload == "knn" else bench_cifar(args, world, rank, dev)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1 or args.rccl_world1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
