"""`scripts/seeddelta-policy-eval.sh` as one driver: build -> greedy smoke -> perplexity
base vs SeedDelta, per context length, with the script's directory layout.

The reference script (scripts/seeddelta-policy-eval.sh:195-487) shells out to three binaries:
`llama-seeddelta-build` (here: build.build_gguf, the GPU builder), `llama-cli` for a five-prompt greedy
battery and `llama-perplexity`, both with and without `--seeddelta` (:353-469). The runtime binaries are
the reference's own programs linked against a libllama whose src/llama-seeddelta.cpp is replaced by the
shim; this module only needs a `runner` command that evaluates a GGUF on a token
stream and prints one JSON line {"ppl", "greedy", ...}. In this repository that is
oracle/_ref/sd_e2e_shim (the reference's libllama + the shim, driven through include/llama.h by
oracle/e2e_driver.cpp) because tools/perplexity and tools/cli cannot be built without the reference's
cmake-generated sources; with a full shim-linked reference build `runner` points at a wrapper around
llama-perplexity / llama-cli instead.

Files written under `outdir`, named as the script names them (:188-193, :438-451):
  model_sd.gguf, report.json, build.log,
  ctx<N>/perplexity_base.log, ctx<N>/perplexity_seeddelta.log   (each ends with the line
      "Final estimate: PPL = <x> +/- 0" that scripts/seeddelta-autogate.py:340-347 parses),
  ctx<N>/greedy_<i>_base.txt, greedy_<i>_seeddelta.txt, greedy_<i>.diff,
  summary.json (this driver's own digest: PPL pair, relative delta, greedy agreement per prompt).
"""
from __future__ import annotations

import difflib
import json
import os
import subprocess
from typing import List, Optional, Sequence

from . import scan as SC


def run_runner(runner: Sequence[str], model: str, tokens: str, ctx: int, batch: int, chunks: int, threads: int,
               seeddelta: bool, greedy: Optional[tuple] = None, logits_out: str = "", env: Optional[dict] = None) -> dict:
    cmd = list(runner) + [model, tokens, "--ctx", str(ctx), "--batch", str(batch), "--chunks", str(chunks),
                          "--threads", str(threads)]
    if seeddelta:
        cmd.append("--seeddelta")
    if greedy:
        cmd += ["--greedy", str(greedy[0]), str(greedy[1])]
    if logits_out:
        cmd += ["--logits-out", logits_out]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env)
    if p.returncode != 0:
        raise RuntimeError(f"runner failed ({p.returncode}): {' '.join(cmd)}\n{p.stderr[-2000:]}")
    out = json.loads(p.stdout.strip().splitlines()[-1])
    out["cmd"] = cmd
    out["stderr"] = p.stderr
    return out


def policy_eval(base_model: str, tokens: str, outdir: str, runner: Sequence[str], *, params: SC.ScanParams,
                layers: str = "", policy_file: str = "", imatrix_file: str = "", strip_dense: bool = True,
                ctx_list: Sequence[int] = (512,), chunks: int = 16, n_predict: int = 256, threads: int = 8,
                batch: Optional[int] = None, prompts: Sequence[int] = (4, 8, 12, 16, 24),
                base_runner: Optional[Sequence[str]] = None, model_sd: str = "", eval_only: bool = False,
                build_only: bool = False, K_gate: int = 0, K_up: int = 0, K_down: int = 0) -> dict:
    """prompts: prompt lengths (tokens taken from the head of the token file) of the greedy battery — the
    script's five fixed text prompts (:401-407) need a tokenizer, which the synthetic models here lack.
    base_runner: the command for the runs without --seeddelta (default: `runner`)."""
    from .build import build_gguf
    os.makedirs(outdir, exist_ok=True)
    sd_model = model_sd or os.path.join(outdir, "model_sd.gguf")
    report_json = os.path.join(outdir, "report.json")
    summary: dict = {"base": base_model, "model_sd": sd_model, "ctx": {}}
    if not eval_only:
        res = build_gguf(base_model, sd_model, params, layers=layers, policy_file=policy_file, imatrix_file=imatrix_file,
                         strip_dense=strip_dense, report_json=report_json, K_gate=K_gate, K_up=K_up, K_down=K_down)
        with open(os.path.join(outdir, "build.log"), "w") as fh:
            fh.write("== build ==\n")
            for e in res["entries"]:
                fh.write(json.dumps({k: e.get(k) for k in ("layer", "kind", "K", "emit", "gating_pass", "strip_dense")}) + "\n")
        summary["build"] = {"n_added": res["n_added"], "stripped": res["stripped"],
                            "emitted": [(e["layer"], e["kind"]) for e in res["entries"] if e.get("emit")]}
    if build_only:
        return summary
    base_runner = list(base_runner or runner)
    for ctx in ctx_list:
        d = os.path.join(outdir, f"ctx{ctx}")
        os.makedirs(d, exist_ok=True)
        b = batch or ctx
        entry: dict = {"greedy": []}
        # greedy smoke battery (:409-416)
        for i, plen in enumerate(prompts):
            gb = run_runner(base_runner, base_model, tokens, ctx, b, 0, threads, False, greedy=(plen, n_predict))
            gs = run_runner(runner, sd_model, tokens, ctx, b, 0, threads, True, greedy=(plen, n_predict))
            tb, ts = " ".join(map(str, gb["greedy"])), " ".join(map(str, gs["greedy"]))
            for tag, g, txt in (("base", gb, tb), ("seeddelta", gs, ts)):
                with open(os.path.join(d, f"greedy_{i}_{tag}.txt"), "w") as fh:
                    fh.write("CMD: " + " ".join(g["cmd"]) + "\n----\n" + txt + "\n")
            with open(os.path.join(d, f"greedy_{i}.diff"), "w") as fh:
                fh.writelines(difflib.unified_diff([tb + "\n"], [ts + "\n"], f"greedy_{i}_base.txt", f"greedy_{i}_seeddelta.txt"))
            same = 0
            for x, y in zip(gb["greedy"], gs["greedy"]):
                if x != y:
                    break
                same += 1
            entry["greedy"].append({"prompt_tokens": plen, "n_predict": n_predict, "common_prefix": same,
                                    "identical": gb["greedy"] == gs["greedy"], "seeddelta_tokens": gs["greedy"]})
        # perplexity (:418-420)
        pb = run_runner(base_runner, base_model, tokens, ctx, b, chunks, threads, False)
        ps = run_runner(runner, sd_model, tokens, ctx, b, chunks, threads, True)
        for tag, r in (("base", pb), ("seeddelta", ps)):
            with open(os.path.join(d, f"perplexity_{tag}.log"), "w") as fh:
                fh.write("CMD: " + " ".join(r["cmd"]) + "\n----\n" + r["stderr"])
                fh.write("".join(f"[{k + 1}]{v:.4f}," for k, v in enumerate(r["chunk_ppl"])) + "\n")
                fh.write(f"Final estimate: PPL = {r['ppl']:.4f} +/- 0\n")
        entry.update(ppl_base=pb["ppl"], ppl_seeddelta=ps["ppl"], delta_ppl_rel=(ps["ppl"] - pb["ppl"]) / pb["ppl"],
                     chunks=ps["chunks"], count=ps["count"])
        summary["ctx"][str(ctx)] = entry
    with open(os.path.join(outdir, "summary.json"), "w") as fh:
        json.dump(summary, fh, indent=2)
    return summary


def parse_ppl(path: str) -> Optional[float]:
    """scripts/seeddelta-autogate.py:340-347."""
    try:
        with open(path) as fh:
            for line in fh:
                if line.startswith("Final estimate: PPL ="):
                    return float(line.split("=")[1].split("+/-")[0].strip())
    except OSError:
        return None
    return None
