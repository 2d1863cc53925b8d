"""Shared set-up of the reference's embedding CLIs (run_lemon.py:59-131 and its twin
lib/baselines/discrepancy_baseline.py:52-117 repeat the same block): output dir + Tee, device / process group, seeding,
args.json, model + tokenizer, datasets, the DB-subset draw, prompt construction and the per-split embedding pass."""
import json
import os
import random
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

CLF_DATASETS = ["cifar10", "cifar100", "cifar10_full", "cifar100_full", "mini_imagenet", "stanford_cars"]


class Tee:
    """lib/utils/utils.py:42-54"""

    def __init__(self, fname, stream, mode="a"):
        self.stream, self.file = stream, open(fname, mode)

    def write(self, m):
        self.stream.write(m); self.file.write(m); self.flush()

    def flush(self):
        self.stream.flush(); self.file.flush()


def run_with_tee(fn, args):
    """The Tee objects are per run (the reference scripts are one-shot; ours are functions)."""
    saved = (sys.stdout, sys.stderr)
    try:
        return fn(args)
    finally:
        for cur in (sys.stdout, sys.stderr):
            if isinstance(cur, Tee):
                cur.file.close()
        sys.stdout, sys.stderr = saved


def add_extension_flags(p):
    """Flags that are not in the reference: local weights / data, encoder batching, caches."""
    p.add_argument("--clip_path", default="random",
                   help="local HF CLIP checkpoint dir (huggingface_clip), local OpenAI-format .pt (in-tree CLIP branches), local copy of the "
                        "BiomedCLIP hub snapshot -- open_clip_pytorch_model.bin + vocab.txt -- (biomed_clip), or random[:arch]")
    p.add_argument("--bpe_path", default=None,
                   help="CLIP BPE merges file (bpe_simple_vocab_16e6.txt.gz / merges.txt) when the checkpoint has no tokenizer files; "
                        "biomed_clip: the BERT vocab.txt when it is not beside the weights")
    p.add_argument("--data_root", default="./data", help="local dataset root, or synthetic:N")
    p.add_argument("--algo", default="auto", choices=["auto", "f32", "bf16"], help="kNN scan algorithm")
    p.add_argument("--encoder_batch", default=512, type=int, help="encoder micro-batch on the GPU")
    p.add_argument("--no_text_dedup", action="store_true",
                   help="encode every sample's prompt (the reference does) instead of each distinct prompt once")
    p.add_argument("--embedding_cache", default=None,
                   help="directory for per-split embedding caches (lemon_amd/cache.py): re-runs with another k / metric / "
                        "ablation skip the encoder")
    return p


def prepare(args):
    hparams = vars(args)
    out_dir = Path(args.output_dir)
    out_dir.mkdir(exist_ok=True, parents=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not args.debug and rank == 0:
        sys.stdout = Tee(os.path.join(args.output_dir, "out.txt"), sys.stdout)
        sys.stderr = Tee(os.path.join(args.output_dir, "err.txt"), sys.stderr)

    from . import _lib, datasets as ds
    from . import clip as clip_mod
    from . import data as data_mod
    from .pipeline import Embedder, shard_bounds

    _lib.load()
    if not torch.cuda.is_available():
        raise _lib.LemonHipError("this command needs a HIP device (no CPU fallback for the hot path)")
    device = torch.device("cuda", local % max(torch.cuda.device_count(), 1))     # rehearsals: ranks may share one card
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LEMON_DIST_BACKEND", "nccl")   # nccl = RCCL over xGMI; gloo only to rehearse on one card
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if rank == 0:
        print("Environment:")
        print("\tPython: {}".format(sys.version.split(" ")[0]))
        print("\tPyTorch: {}".format(torch.__version__))
        print("\tHIP: {}".format(torch.version.hip))
        print("\tNumPy: {}".format(np.__version__))
        print("\tNode: {}".format(socket.gethostname()))
        print("\tDevice: {} x{}".format(torch.cuda.get_device_name(device), world))
        print("Args:")
        for k, v in sorted(hparams.items()):
            print("\t{}: {}".format(k, v))

    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if rank == 0:
        with open(out_dir / "args.json", "w") as f:
            json.dump(vars(args), f, default=str)

    label_set = ds.LABEL_SETS.get(args.dataset)
    is_clf = args.dataset in CLF_DATASETS
    model, tokenizer = clip_mod.algorithm_class_from_scratch(args.clip_model, text_base_name=args.clip_path, img_base=None,
                                                             return_tokenizer=True, bpe_path=args.bpe_path)
    hf_style = args.clip_model == "huggingface_clip"      # the other branches' tokenizer returns a LongTensor (:148-154)
    train_set, val_set, test_set = data_mod.get_dataset(args.dataset, args.data_seed, percent_flips=args.noise_level,
                                                        flip_type=args.noise_type, data_root=args.data_root,
                                                        image_size=model.cfg.image_size)
    if getattr(args, "subset_val_set", -1) > 0:
        rng = np.random.default_rng(args.data_seed)
        val_set = val_set.subset(rng.choice(np.arange(len(val_set)), min(args.subset_val_set, len(val_set)), replace=False))

    # identical prompts are embedded once and gathered (class datasets: C prompts for N samples): the same function
    # of the input, independent of how samples fall into micro-batches; --no_text_dedup encodes every sample's prompt
    embedder = Embedder(model, device, batch_size=args.encoder_batch, text_dedup=not args.no_text_dedup)
    prefix = "A photo of a " if args.custom_cifar_prompt is None else args.custom_cifar_prompt
    prompt_fn = lambda x: prefix + x

    # DB subset: first consumer of the global numpy stream after seeding (run_lemon.py:81,121-127)
    if len(train_set) > args.compr_dataset_size_limit:
        train_indices_in_compr = np.random.choice(np.arange(len(train_set)), args.compr_dataset_size_limit, replace=False)
    else:
        train_indices_in_compr = np.arange(len(train_set))

    def texts_of(noisy, clean):
        if is_clf:
            noisy_txt = label_set[np.asarray(noisy)].tolist()
            clean_txt = label_set[np.asarray(clean)].tolist()
            return noisy_txt, clean_txt, [prompt_fn(t) for t in noisy_txt]
        return list(noisy), list(clean), list(noisy)

    def tokenize(prompts):
        if not hf_style:
            return tokenizer(prompts)
        enc = tokenizer(prompts, padding="max_length", truncation=True)
        return torch.tensor(enc["input_ids"])

    # int ids for the discrete text metric (it compares prompt STRINGS, :266-267).  Class datasets: the
    # prompt is a bijection of the noisy label.  Captions: a dictionary built from the dataset itself,
    # in a fixed order, so every rank derives the same ids.
    text_ids = {}
    if not is_clf:
        for dset in (train_set, val_set, test_set):
            for cap in dset.noisy:
                text_ids.setdefault(cap, len(text_ids))

    def ids_of(meta):
        if is_clf:
            return np.asarray(meta["noisy"], dtype=np.int32)
        return np.array([text_ids[p] for p in meta["prompts"]], dtype=np.int32)

    from .cache import EmbeddingCache
    cache = EmbeddingCache(args.embedding_cache, dataset=args.dataset, noise_type=args.noise_type,
                           noise_level=args.noise_level, data_seed=args.data_seed, clip_model=args.clip_model,
                           clip_path=os.path.abspath(args.clip_path) if os.path.exists(str(args.clip_path)) else args.clip_path,
                           data_root=args.data_root, prompt=args.custom_cifar_prompt,
                           subset_val_set=getattr(args, "subset_val_set", -1))

    def embed_split(dset, sname):
        """this rank's contiguous shard of a split -> (emb_img, emb_txt, meta) on the device"""
        lo, hi = shard_bounds(len(dset), world, rank)
        if cache.root:
            sl = slice(lo, hi)
            key_prompts = texts_of(dset.noisy[sl], dset.clean[sl])[2]
            hit = cache.load(sname, lo, hi, key_prompts, device)
            if hit is not None:
                return hit
        imgs, toks, meta = [], [], dict(noisy=[], clean=[], noisy_txt=[], clean_txt=[], prompts=[])
        # data chunks of the encoder micro-batch (the reference's --batch_size only sizes its DataLoader batches;
        # per-sample results do not depend on it)
        for px, clean, noisy in dset.batches(max(args.batch_size, args.encoder_batch), lo, hi, device=device):
            noisy_txt, clean_txt, prompts = texts_of(noisy, clean)
            imgs.append(embedder.embed_images(px))
            toks.append(tokenize(prompts))
            meta["noisy"] += list(noisy); meta["clean"] += list(clean)
            meta["noisy_txt"] += noisy_txt; meta["clean_txt"] += clean_txt; meta["prompts"] += prompts
        d = embedder.model.cfg.embed_dim
        e_img = torch.cat(imgs) if imgs else torch.empty((0, d), device=device)
        e_txt = embedder.embed_texts(torch.cat(toks)) if toks else torch.empty((0, d), device=device)
        embedder.raise_if_nonfinite()
        if getattr(embedder, "fold_fallback_batches", 0) and not getattr(embedder, "_fold_fallback_reported", 0) == embedder.fold_fallback_batches:
            print(f"note: {getattr(embedder, 'fold_fallback_rows', 0)} sample(s) in {embedder.fold_fallback_batches} encoder micro-batch(es) so far held a row whose "
                  "mean lies beyond the folded LayerNorm's bound and were embedded again with LayerNorm kernels (same arithmetic otherwise)")
            embedder._fold_fallback_reported = embedder.fold_fallback_batches
        if embedder.fallback_batches and not getattr(embedder, "_fallback_reported", 0) == embedder.fallback_batches:
            print(f"note: {getattr(embedder, 'fallback_rows', 0)} sample(s) in {embedder.fallback_batches} encoder micro-batch(es) so far left the fp16 range of the "
                  "split GEMM operands and were embedded again with bf16x6 operands / fp32 attention (same fp32-equivalent arithmetic, no range limit)")
            embedder._fallback_reported = embedder.fallback_batches
        meta["lo"] = lo
        if cache.root:
            cache.store(sname, lo, hi, meta["prompts"], e_img, e_txt, meta)
        return e_img, e_txt, meta

    def gather_meta(meta):
        """(meta, is_mislabel) of a whole split on every rank: per-rank python lists -> global order"""
        flips = 1 - (np.array(meta["noisy_txt"]) == np.array(meta["clean_txt"]))
        if world > 1:
            import torch.distributed as dist
            gathered = [None] * world
            dist.all_gather_object(gathered, (meta, flips))
            meta = {key: sum((g[0][key] for g in gathered), []) for key in ("noisy", "clean", "noisy_txt", "clean_txt")}
            flips = np.concatenate([g[1] for g in gathered])
        return meta, flips

    return SimpleNamespace(out_dir=out_dir, world=world, rank=rank, device=device, label_set=label_set, is_clf=is_clf,
                           model=model, tokenizer=tokenizer, embedder=embedder, prompt_fn=prompt_fn, tokenize=tokenize,
                           sets={"train": train_set, "val": val_set, "test": test_set},
                           train_indices_in_compr=train_indices_in_compr, texts_of=texts_of, ids_of=ids_of,
                           embed_split=embed_split, gather_meta=gather_meta)


def meta_columns(sname, n_total, meta, flips):
    """The label columns of the per-sample record (run_lemon.py:291-299)."""
    return {"sset": sname, "idx": np.arange(n_total),
            "actual_label": [c.item() if hasattr(c, "item") else c for c in meta["clean"]],
            "actual_label_text": meta["clean_txt"], "noisy_label": list(meta["noisy"]),
            "noisy_label_text": meta["noisy_txt"], "is_mislabel": flips, "is_correct_label": 1 - flips}
