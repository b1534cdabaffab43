#!/usr/bin/env python3
"""Static instruction MIX of one kernel PER PHASE of the control step (the phases of the KM_PROFILE stamps), from a
`hipcc -S -gline-tables-only` listing whose .loc comments carry the inlined-at chain -- and, with --weights, a DYNAMIC estimate
(static counts x how often a wave runs the phase per control step), checked against the hardware's SQ_INSTS_VALU.

An instruction belongs to the phase of the OUTERMOST frames of its chain: the call site in k_step (before_step / step1_products /
solve / integrate / tail ...), refined by the call site in step1_products, solve_newton_sl and newton_loop_sl (the latter split
at its pf.ph() stamps: H build | factorisation | solves | line-search set-up | line search | evaluation).  Helpers inlined
everywhere (dppfma*, gsum_n, frcp, ...) are thereby charged to the phase that called them, which per-function tables cannot do.

Usage: tools/asm_phase_mix.py <listing.s> <kernel-symbol-substring> [--src gym_kmanip_amd/csrc/kmanip_dyn.hip] [--weights k=v,...]
"""
import collections
import os
import re
import sys

CLASSES = ["f64", "mov", "dppmov", "sel", "agpr", "lane", "cmp", "valu", "lds", "vmem", "salu", "nop", "wait", "br"]
VALU = ["f64", "mov", "dppmov", "sel", "agpr", "lane", "cmp", "valu"]


def klass(op, s):
    if op.startswith("v_accvgpr"):
        return "agpr"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith("v_cndmask"):
        return "sel"
    if op.startswith(("v_mov", "v_pk_mov")):
        return "dppmov" if ("dpp" in op or "row_" in s or "quad_perm" in s) else "mov"
    if op.startswith(("v_permlane", "v_swap")):
        return "dppmov"
    if op.startswith("v_cmp"):
        return "cmp"
    if "f64" in op:
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    return "salu"


def newton_stamps(src):
    """line numbers of newton_loop_sl's stamps: {slot offset: [lines]} for pf.ph(<k> + 6 * S)"""
    out = collections.defaultdict(list)
    for i, ln in enumerate(src, 1):
        m = re.search(r"pf\.ph\((\d+) \+ 6 \* S\)", ln)
        if m:
            out[int(m.group(1))].append(i)
    return out


def phase_of(chain, src, stamps, fn_of):
    """chain: [(file, line)] innermost first.  Returns the phase name."""
    frames = [(f, l) for f, l in reversed(chain) if f == "kmanip_dyn.hip"]           # outermost first
    top = None
    loop = None
    for f, l in frames:
        text = src[l - 1] if 0 < l <= len(src) else ""
        fn = fn_of(l)
        if fn == "k_step":
            if "coop_before_step" in text:
                # refined by where in coop_trf (kmanip_ik_coop.hpp) the instruction sits: set-up | once per outer iteration
                # (normal matrix, scaling) | once per trial point (trust-region solve, step, evaluation, ratio) | acceptance tests
                for f2, l2 in reversed(chain):
                    if f2 == "kmanip_ik_coop.hpp" and IK["lo"] <= l2 <= IK["hi"]:
                        if l2 < IK["outer"]: return "IK: trf set-up"
                        if l2 < IK["trial"]: return "IK: per outer iteration"
                        if l2 <= IK["trial_end"]: return "IK: per trial point"
                        return "IK: acceptance / tests"
                return "IK: decode, first evaluation, write-back"
            if "step1_products" in text: top = "step1"; continue
            if "solve<" in text: top = "solve"; continue
            if "integrate<" in text: return "integrate"
            if "load_state" in text or "init_ws" in text or "stage_model" in text: return "load/stage"
            if "store_state" in text: return "tail"
            if "reset_env" in text: return "reset"
            if any(k in text for k in ("fk_parallel", "collide_parallel", "env_reward", "write_obs")): return "tail"
            return "k_step glue"
        if fn == "step1_products":
            for key, name in (("fk_parallel", "fk"), ("bias_bodies", "bias bodies"), ("cube_bias", "bias bodies"), ("collide_parallel", "collide"),
                              ("composite_", "composite+M+bias"), ("mass_matrix", "composite+M+bias"), ("bias_project", "composite+M+bias"),
                              ("invert_mass", "invert_mass"), ("build_constraints", "build_constraints")):
                if key in text: return name
            return "step1 glue"
        if fn == "solve_newton":
            if "solve_newton_sl" in text: continue
            return "newton a_s"
        if fn == "solve_newton_sl":
            if "newton_loop_sl<NL, G, KM_SUB_ALL, true>" in text: loop = "JOINT"; continue
            if "newton_loop_sl<NL, G, KM_SUB_ARM>" in text: loop = "ARM"; continue
            if "newton_loop_sl<NL, G, KM_SUB_CUBE>" in text: loop = "CUBE"; continue
            if "newton_loop_sl<NL, G, KM_SUB_ALL>" in text: loop = "ALL2"; continue
            return "newton start evals"
        if fn == "newton_loop_sl" and loop:
            # stamps in source order: 9 (after H build), 10 (after factorisation; several sites), 11 (after the solves / the plain
            # direction), 12 (after ls set-up), 13 (after the line search), 14 (after the evaluation)
            s9, s10, s11, s12, s13, s14 = (stamps[k] for k in (9, 10, 11, 12, 13, 14))
            if l <= min(s11): sub = "plain direction / entry"
            elif l <= max(s9): sub = "H build"
            elif l <= max(s10): sub = "factorisation"
            elif l <= max(s11): sub = "solves"
            elif l <= max(s12): sub = "ls set-up"
            elif l <= max(s13): sub = "line search"
            else: sub = "evaluation"
            return "%s: %s" % (loop, sub)
        if fn == "solve": continue
    if top == "solve":
        return "solve glue"
    return top or "?"


IK = {}


def ik_lines(path):
    """line landmarks of coop_trf in kmanip_ik_coop.hpp: its extent, the outer `for (;;)`, the inner `while (actual <= 0 ...)`, stamp 37"""
    src = open(path).read().split("\n")
    lo = next(i for i, l in enumerate(src, 1) if "int coop_trf(" in l)
    hi = next(i for i, l in enumerate(src, 1) if i > lo and l.startswith("}"))
    IK.update(lo=lo, hi=hi,
              outer=next(i for i, l in enumerate(src, 1) if i > lo and "for (;;)" in l),
              trial=next(i for i, l in enumerate(src, 1) if i > lo and "while (actual <= 0" in l),
              trial_end=next(i for i, l in enumerate(src, 1) if i > lo and "pf->ph(37)" in l))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path, sym = args[0], args[1]
    opt = dict(a[2:].split("=", 1) if "=" in a else (a[2:], "1") for a in sys.argv[1:] if a.startswith("--"))
    here = os.path.dirname(os.path.abspath(__file__))
    src_path = opt.get("src", os.path.join(here, "..", "gym_kmanip_amd", "csrc", "kmanip_dyn.hip"))
    src = open(src_path).read().split("\n")
    stamps = newton_stamps(src)
    ik_lines(os.path.join(os.path.dirname(src_path), "kmanip_ik_coop.hpp"))
    # function of a source line: the nearest preceding "__device__ ... name(" / "__global__ ... name(" definition
    starts = []
    for i, ln in enumerate(src, 1):
        if not re.match(r"^(?:__device__|__global__|static)\b", ln) or ln.startswith("static_assert") or ln.rstrip().endswith(";"):
            continue
        names = [n for n in re.findall(r"\b([A-Za-z_0-9]+)\s*\(", ln) if n not in ("__launch_bounds__", "__attribute__")]
        if names:
            starts.append((i, names[0]))

    def fn_of(line):
        name = None
        for i, n in starts:
            if i > line:
                break
            name = n
        return name

    counts = collections.defaultdict(collections.Counter)
    inside = False
    chain = None
    for ln in open(path, errors="replace"):
        s = ln.strip()
        if re.match(r"^[_A-Za-z0-9.$]+:", s) and not s.startswith("."):
            inside = sym in s
            continue
        if not inside:
            continue
        if s.startswith(".loc"):
            c = s.split(";", 1)[1] if ";" in s else ""
            chain = [(m.group(1).split("/")[-1], int(m.group(2))) for m in re.finditer(r"(\S+?):(\d+):\d+", c)]
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        ph = phase_of(chain or [], src, stamps, fn_of)
        counts[ph][klass(op, s)] += 1

    weights = {}
    for kv in opt.get("weights", "").split(","):
        if "=" in kv:
            k, v = kv.split("=")
            weights[k.strip()] = float(v)

    def w_of(ph):
        for k, v in weights.items():
            if ph.startswith(k):
                return v
        return None

    order = sorted(counts, key=lambda p: -sum(counts[p].values()))
    print("%-34s %6s | %s | %6s %6s" % ("phase (static instructions)", "total", " ".join("%6s" % c for c in CLASSES), "VALU", "non-f64"))
    tot = collections.Counter()
    for ph in order:
        c = counts[ph]
        tot.update(c)
        valu = sum(c[k] for k in VALU)
        print("%-34s %6d | %s | %6d %6d" % (ph[:34], sum(c.values()), " ".join("%6d" % c[k] for k in CLASSES), valu, valu - c["f64"]))
    valu = sum(tot[k] for k in VALU)
    print("%-34s %6d | %s | %6d %6d" % ("TOTAL", sum(tot.values()), " ".join("%6d" % tot[k] for k in CLASSES), valu, valu - tot["f64"]))
    if weights:
        print()
        print("dynamic estimate per wave and control step = static x executions (--weights; phases without a weight are left out):")
        print("%-34s %6s | %8s %8s %8s %8s %8s %8s %8s %8s | %8s" % ("phase", "x", "f64", "mov", "dppmov", "sel", "agpr", "lane", "cmp", "valu", "non-f64"))
        dyn = collections.Counter()
        rows = []
        for ph in order:
            x = w_of(ph)
            if x is None:
                continue
            c = counts[ph]
            rows.append((sum(c[k] for k in VALU) * x, ph, x, c))
        for _, ph, x, c in sorted(rows, reverse=True):
            for k in VALU:
                dyn[k] += c[k] * x
            print("%-34s %6.1f | %s | %8.0f" % (ph[:34], x, " ".join("%8.0f" % (c[k] * x) for k in VALU), sum(c[k] for k in VALU if k != "f64") * x))
        tv = sum(dyn.values())
        print("%-34s %6s | %s | %8.0f" % ("SUM", "", " ".join("%8.0f" % dyn[k] for k in VALU), tv - dyn["f64"]))
        print("VALU instructions per wave and control step, estimated: %.0f (f64 arithmetic %.0f = %.1f %%)" % (tv, dyn["f64"], 100 * dyn["f64"] / max(tv, 1)))


if __name__ == "__main__":
    main()
