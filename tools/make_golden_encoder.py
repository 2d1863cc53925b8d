"""Encoder goldens from the reference's in-tree CLIP (lib/models/chexzero_clip.py:263-392 CLIP, :458-479 load_clip),
imported in the build container with `ftfy` stubbed (only simple_tokenizer.py needs it).  Weights and inputs come from the
seeded recipe in tests/encoder_recipe.py, which the tests re-run; stored: the reference's encode_image / encode_text
outputs (+ checksums of weights and inputs to catch RNG drift).  -> tests/golden/encoder_chexzero.npz
Run:  python tools/make_golden_encoder.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
ftfy = types.ModuleType("ftfy")
ftfy.fix_text = lambda t: t
sys.modules["ftfy"] = ftfy
from lib.models import chexzero_clip as cz          # noqa: E402
from tests.encoder_recipe import CONFIGS, inputs, openai_state_dict      # noqa: E402

out = {}
for name, cfg in CONFIGS.items():
    torch.manual_seed(0)
    if name == "scratch_b16_77":
        model = cz.load_clip(None, context_length=77)           # the reference's own constructor path (:458-479)
        assert model.context_length == 77 and model.visual.conv1.weight.shape == (768, 3, 16, 16)
    elif name == "scratch_b16_256":
        model = cz.load_clip(None)                              # default context_length (256): the MIMIC configuration
        assert model.context_length == 256 and model.positional_embedding.shape == (256, 512)
    else:
        model = cz.CLIP(**cfg)
    sd = openai_state_dict(cfg)
    missing = model.load_state_dict(sd, strict=True)
    model = model.float().eval()
    px, ids = inputs(cfg)
    with torch.no_grad():
        img = model.encode_image(px)
        txt = model.encode_text(ids)
    out[f"{name}_img"], out[f"{name}_txt"] = img.numpy(), txt.numpy()
    out[f"{name}_weights_abs_sum"] = np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))
    out[f"{name}_inputs_abs_sum"] = np.float64(float(px.double().abs().sum()) + float(ids.double().sum()))
    print(name, "img", tuple(img.shape), float(img.abs().mean()), "txt", tuple(txt.shape), float(txt.abs().mean()), flush=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_chexzero.npz"), **out)
