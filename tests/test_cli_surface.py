"""CPU: the CLI keeps run_lemon.py's argument surface (run_lemon.py:35-57): names, defaults, choices."""
import pytest

from lemon_amd.run_lemon import build_parser

REFERENCE_FLAGS = {   # name: (default, choices or None) as at run_lemon.py:35-57
    "exp_name": (None, None), "output_dir": (None, None),
    "dataset": ("cifar100", ["cifar10", "cifar100", "flickr30k", "mscoco", "mimiccxr_caption", "mmimdb", "cifar10_full",
                             "cifar100_full", "mini_imagenet", "stanford_cars", "cc3m"]),
    "noise_type": ("real", ["real", "asymmetric", "symmetric", "random", "noun", "cat"]),
    "noise_level": (0.4, None), "dist_type": ("cosine", ["cosine", "euclidean"]), "normalize_d1": (False, None),
    "clip_model": ("huggingface_clip", ["huggingface_clip", "biomed_clip", "mimic_clip_from_scratch_random",
                                        "mimic_clip_from_scratch_cat", "chexzero", "cc3m_clip_from_scratch"]),
    "knn_k": (5, None), "batch_size": (128, None), "seed": (0, None), "data_seed": (0, None),
    "compr_dataset_size_limit": (50000, None),
    "ablation": ("none", ["none", "tau_1", "tau_2", "tau_1_2", "beta", "gamma", "multimodal_baseline", "d1",
                          "only_gamma", "only_beta"]),
    "use_discrete_for_text": (False, None), "real_dataset": (False, None), "custom_cifar_prompt": (None, None),
    "subset_val_set": (-1, None), "debug": (False, None), "skip_train": (False, None), "skip_hparam_optim": (False, None),
}


def test_all_21_reference_flags_present_with_same_defaults_and_choices():
    p = build_parser()
    acts = {a.dest: a for a in p._actions}
    assert len(REFERENCE_FLAGS) == 21
    for name, (default, choices) in REFERENCE_FLAGS.items():
        assert name in acts, name
        assert acts[name].default == default, name
        if choices is not None:
            assert list(acts[name].choices) == choices, name
    assert acts["output_dir"].required


def test_cat_noise_passes_argparse_like_upstream():
    a = build_parser().parse_args(["--output_dir", "o", "--dataset", "cifar100", "--noise_type", "cat"])
    assert a.noise_type == "cat"       # fails later, in add_noisy_labels, exactly like the reference
    with pytest.raises(SystemExit):
        build_parser().parse_args(["--output_dir", "o", "--dist_type", "manhattan"])


def test_generic_transform_shape_and_normalisation():
    import numpy as np
    from PIL import Image
    from lemon_amd.data import generic_transform
    from lemon_amd import datasets as ds
    img = Image.fromarray(np.full((32, 32, 3), 255, np.uint8))
    x = generic_transform(img)
    assert tuple(x.shape) == (3, 224, 224)
    for c in range(3):
        assert abs(float(x[c].mean()) - (1 - ds.CLIP_MEAN[c]) / ds.CLIP_STD[c]) < 1e-5
    wide = Image.fromarray(np.zeros((100, 300, 3), np.uint8))
    assert tuple(generic_transform(wide).shape) == (3, 224, 224)
