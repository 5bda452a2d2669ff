#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.json by EXECUTING the reference's own importable files.

Run in the build container only (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden_from_reference.py
Importable pieces of the hot path: the action tokenizer package
`src/tokenizers_custom` (MinMaxTokenizer with shift, used at multi_domain_discrete_dt_model.py:56-60)
and `src/algos/models/rms_norm.py` (LlamaRMSNorm, used at decision_xlstm.py:190-191).  Everything else on
the path needs gym / stable_baselines3 / xlstm / mamba_ssm, which are not installable here.
Only inputs and outputs are stored -- no reference source text.
"""
import importlib.util
import json
import os
import sys

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _exec_pieces(path, names):
    """Execute only the named top-level assignments / functions of a reference file (its imports need robosuite /
    dmc2gym, which are absent): the observation-space tables and `map_flattened_obs_to_full_space`."""
    import ast
    import numpy as np
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body
            if (isinstance(n, ast.Assign) and any(getattr(t, "id", None) in names for t in n.targets))
            or (isinstance(n, ast.FunctionDef) and n.name in names)]
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def obs_full_space_vectors(g):
    """Inputs / outputs of the reference's flattened-observation -> full-space mapping for one DMControl and one
    Mimicgen observation spec (src/envs/dmcontrol_utils.py:35-78, src/envs/mimicgen_utils.py:58-78,199-214)."""
    import numpy as np
    out = {}
    cases = {
        "dmc_cheetah_run": ("src/envs/dmcontrol_utils.py", ["DMC_OBSTYPE_TO_DIM", "DMC_FULL_OBS_DIM", "DMC_OBSTYPE_TO_STARTIDX"],
                            [("position", 8), ("velocity", 9)]),
        "dmc_walker_walk": ("src/envs/dmcontrol_utils.py", ["DMC_OBSTYPE_TO_DIM", "DMC_FULL_OBS_DIM", "DMC_OBSTYPE_TO_STARTIDX"],
                            [("orientations", 14), ("height", 1), ("velocity", 9)]),
        "mimicgen_main_lowdim": ("src/envs/mimicgen_utils.py",
                                 ["MIMICGEN_OBSTYPE_TO_DIM", "MIMICGEN_FULL_OBS_DIM", "MIMICGEN_OBSTYPE_TO_STARTIDX"],
                                 [("robot0_eef_pos", 3), ("robot0_eef_quat", 4), ("robot0_gripper_qpos", 2), ("object", 23)]),
    }
    for name, (rel, tables, spec) in cases.items():
        ns = _exec_pieces(os.path.join(REF, rel), tables + ["map_flattened_obs_to_full_space"])
        n = sum(d for _, d in spec)
        x = (torch.rand(4, n, generator=g) * 2 - 1).numpy().astype(np.float32)
        obs_spec = {k: np.zeros(d, dtype=np.float32) for k, d in spec}   # the function reads .shape of each entry
        full = ns["map_flattened_obs_to_full_space"](x, obs_spec)
        dims = ns[tables[0]]
        out[name] = {"spec": [[k, d] for k, d in spec], "x": x.tolist(), "full": np.asarray(full, dtype=np.float32).tolist(),
                     "full_dim": int(ns[tables[1]]), "obstype_to_dim": {k: int(v) for k, v in dims.items()}}
    return out


def main():
    sys.path.insert(0, REF)
    from src.tokenizers_custom import make_tokenizer  # reference code, executed not copied
    out = {"generator": "tests/golden/make_golden_from_reference.py", "reference": "ml-jku/LRAM @ 2024-11-01"}

    tok = make_tokenizer("minmax", {"vocab_size": 256, "shift": 18})
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.tensor([-1.0, -0.999, -0.5, 0.0, 0.5, 0.9921875, 0.9999, 1.0, -1.5, 1.5, 1e-9, -1e-9]),
                   torch.rand(116, generator=g) * 2 - 1]).reshape(16, 8)
    tokens = tok.tokenize(x.clone())
    all_tokens = torch.arange(0, 274).reshape(1, -1)
    inv = tok.inv_tokenize(all_tokens.clone())
    out["minmax_shift18"] = {"x": x.tolist(), "tokens": tokens.tolist(), "inv_table": inv.reshape(-1).tolist()}

    tok0 = make_tokenizer("minmax", {"vocab_size": 256})
    out["minmax_shift0"] = {"x": x.tolist(), "tokens": tok0.tokenize(x.clone()).tolist(),
                            "inv_table": tok0.inv_tokenize(torch.arange(0, 256).reshape(1, -1)).reshape(-1).tolist()}

    spec = importlib.util.spec_from_file_location("ref_rms_norm", os.path.join(REF, "src/algos/models/rms_norm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases = []
    for eps in (1e-6, 1e-5):
        norm = mod.LlamaRMSNorm(48, eps=eps)
        with torch.no_grad():
            norm.weight.copy_(torch.randn(48, generator=g) * 0.3 + 1.0)
        xin = torch.randn(5, 3, 48, generator=g) * 2.0
        with torch.no_grad():
            y = norm(xin)
        cases.append({"eps": eps, "weight": norm.weight.tolist(), "x": xin.tolist(), "y": y.tolist()})
    out["llama_rms_norm"] = cases

    out["obs_full_space"] = obs_full_space_vectors(g)

    with open(os.path.join(HERE, "reference_vectors.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote", os.path.join(HERE, "reference_vectors.json"))


if __name__ == "__main__":
    main()
