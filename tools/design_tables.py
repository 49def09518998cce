#!/usr/bin/env python
"""Rewrites the number rows of DESIGN.md §5 (the contract line table and the per-callback table) from the files under profiles/
that they cite — profiles/r5_bench_default.json and profiles/r5_callbacks_config{2,3,4}.json — so that a refresh of the evidence
(tools/refresh_profiles_r5.sh) and the document cannot drift apart.  usage: python tools/design_tables.py  (from the repo root)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
b = json.loads(open(os.path.join(P, "r5_bench_default.json")).read().strip().splitlines()[-1])
cb = {c: json.load(open(os.path.join(P, f"r5_callbacks_config{c}.json")))["callbacks"] for c in (2, 3, 4)}
path = os.path.join(ROOT, "DESIGN.md")
lines = open(path).read().split("\n")


def sub(prefix, new):
    hits = [i for i, l in enumerate(lines) if l.startswith(prefix)]
    assert hits, prefix
    lines[hits[0]] = new


def g(d):
    return d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["kernel"]


ms, v, f, t, k = g(b)
one, allc = b["cpu_baseline"]["value"], b["cpu_baseline"]["all_cores"]["value"]
sub("| 2 (headline) |", f"| 2 (headline) | LV N = 1e7 | `{k}` | {ms:.4f} | {v:.3g} | 880 MB | {t / 1e6:.0f} MB (×{t / 880e6:.2f}) | **{f:.3f}** | {one:.2g} / {allc:.2g} nnz/s | {v / one:,.0f}× |".replace(",", " "))
d = b["config3"]; ms, v, f, t, k = g(d); one, allc = d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"]
sub("| 3 | rocket", f"| 3 | rocket nh = 1e6 | `{k}` | {ms:.4f} | {v:.3g} | 528 MB | {t / 1e6:.0f} MB | {f:.3f} | {one:.3g} / {allc:.2g} | {v / one:,.0f}× |")
d = b["config4"]; ms, v, f, t, k = g(d); one, allc = d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"]
sub("| 4 | ACOPF", f"| 4 | ACOPF 78 484 buses (synthetic) | `{k}` | {ms:.4f} | {v:.3g} | 102 MB | {t / 1e6:.0f} MB (MALL-resident) | mall: {ms / d['roofline']['launch_floor_ms']:.1f}× the launch floor ({f:.2f} of the HBM peak as a cache rate) | {one:.2g} / {allc:.2g} | {v / one:,.0f}× |")
d = b["config5_n1"]; ms, v, f, t, k = g(d)
tnote = ""
if t is None:       # the PMC passes ran a process whose exa_tune picked the sibling chained kernel (they are within 1 % of each other): say so
    tj = json.load(open(os.path.join(P, "r5_traffic_config5.json")))
    by = tj.get("hbm_bytes_per_launch_by_kernel") or {}
    t, tnote = (by[k], "") if k in by else (tj["hbm_bytes_per_launch"], f" (measured on `{tj['kernel']}`)")
sub("| 5 at N = 1", f"| 5 at N = 1 (`scale_base`) | LV N = 1e8 | `{k}` | {ms:.3f} | {v:.3g} | 8.80 GB | {t / 1e9:.2f} GB{tnote} | {f:.3f} (0.66–0.73 across boxes of the pool) | — | — |")


def c(key, cfg):
    return "{:.4f}".format(cb[cfg][key]["ms"])


def fr(key):
    f = cb[2][key]["frac_of_8TBps"]          # None: inputs + outputs fit the Infinity Cache (no HBM fraction for such a launch)
    return "mall" if f is None else "{:.2f}".format(f)


sub("| `hess_coord!` |", f"| `hess_coord!` | `exa_hess` / `exa_hesscl` / `exa_hessc` | HBM stores | {b['ms_per_step']:.4f} (bench) | {b['roofline']['frac']:.2f} | {b['config3']['ms_per_step']:.4f} | {b['config4']['ms_per_step']:.4f} |")
sub("| `jac_coord!` |", f"| `jac_coord!` | `exa_jac` | instruction issue | {c('jac', 2)} | {fr('jac')} | {c('jac', 3)} | {c('jac', 4)} |")
sub("| `cons_nln!` |", f"| `cons_nln!` | `exa_cons` / `exa_cons1` | instruction issue | {c('cons', 2)} | {fr('cons')} | {c('cons', 3)} | {c('cons', 4)} |")
sub("| `grad!` |", f"| `grad!` | `exa_grad_pull` / `exa_grad` | HBM | {c('grad', 2)} | {fr('grad')} | {c('grad', 3)} | {c('grad', 4)} (zero-fill kernel + a 2 µs kernel: launch latency) |")
sub("| `obj` |", f"| `obj` | `exa_obj` (+ `exa_reduce_partials` beyond 512 workgroups) | HBM / launch | {c('obj', 2)} | {fr('obj')} | {c('obj', 3)} | {c('obj', 4)} |")
sub("| `jprod_nln!` |", f"| `jprod_nln!` | `exa_jprod1` | instruction issue | {c('jprod', 2)} | {fr('jprod')} | {c('jprod', 3)} | {c('jprod', 4)} |")
sub("| `jtprod_nln!` |", f"| `jtprod_nln!` | `exa_jtprodw` (windows) / `exa_jtprod` (atomics) | instruction issue | **{c('jtprod', 2)}** (r2: 0.095) | {fr('jtprod')} | {c('jtprod', 3)} (r2: 0.0615) | {c('jtprod', 4)} (atomics) |")
sub("| `hprod!` |", f"| `hprod!` | `exa_hprodw` / `exa_hprod` | instruction issue | **{c('hprod', 2)}** (r2: 0.172) | {fr('hprod')} | {c('hprod', 3)} (windows; atomics 0.067, r2 0.070) | {c('hprod', 4)} (atomics) |")
sub("| fused obj+cons+jac+hess |", f"| fused obj+cons+jac+hess | `exa_fused` | HBM stores | {c('fused', 2)} | {fr('fused')} | {c('fused', 3)} | {c('fused', 4)} |")
sub("| all five (`exa_eval_all`) |", f"| all five (`exa_eval_all`) | `exa_fused` (+ gradient tiles) | HBM stores | {c('eval_all', 2)} (interleaved block order by default since round 5) | {fr('eval_all')} | {c('eval_all', 3)} | {c('eval_all', 4)} |")
sub("| compressed Hessian |", f"| compressed Hessian | `exa_chessw` (+`s`,`x`) / `exa_chessm` | instruction issue | {c('chess', 2)} | {fr('chess')} | {c('chess', 3)} | {c('chess', 4)} |")
sub("| compressed Jacobian |", f"| compressed Jacobian | `exa_cjacw` / `exa_cjacp` | instruction issue | {c('cjac', 2)} | {fr('cjac')} | {c('cjac', 3)} | {c('cjac', 4)} |")
open(path, "w").write("\n".join(lines))
print("DESIGN.md §5 rows rewritten from profiles/r5_bench_default.json and profiles/r5_callbacks_config{2,3,4}.json")
