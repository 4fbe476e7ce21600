#!/usr/bin/env python3
"""Static VALU instruction mix of the sub-step loop of the bench kernel (kf_lean<2,2,false,SIG_VAS_CIR_E,2>): compiles
csrc/kf_lean.hip to gfx950 assembly and counts the instruction classes between the innermost loop header that contains the
Philox multiplies and its back-branch.  Output (JSON on stdout): per PATH and sub-step (the loop body handles 2 paths per lane).
The loop holds one rarely taken side block per path (the guarded root of a uniform that rounds to 1, mcx_device.h pair_from_words);
the region counted ends at the first back-branch, so one of the two radius roots of the common path may fall outside it: the
class split can be short of the executed one by up to half an rsq + two f64 instructions per path (< 1 % of the modelled time;
the TOTAL the model uses is the measured SQ_INSTS_VALU, not this count).
   python tools/asm_mix.py"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "montecarlo-risk-engine_amd", "csrc")


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "kf_lean.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-DMCX_LEAN_ONE_SIG",
                               "--cuda-device-only", "-S", "kf_lean.hip", "-o", out], cwd=CS, stderr=subprocess.DEVNULL)
        text = open(out).read()
    m = re.search(r"^(_ZN\S*kf_leanILi2ELi2ELb0ELi1ELi2ELb1E[^\s:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M)
    lines = m.group(2).split("\n")
    # innermost loop with v_mad_u64_u32: label .LBBx_y ... s_cbranch* .LBBx_y
    best = None
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if l.startswith(".LBB")}
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            body = lines[labels[mm.group(1)]:i + 1]
            n_mad = sum("v_mad_u64_u32" in b for b in body)
            if n_mad >= 20 and (best is None or len(body) < len(best)):
                best = body
    c = collections.Counter()
    for l in best:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith("."):
            continue
        op = l.split()[0]
        if op.startswith("v_"):
            if "readlane" in op or "writelane" in op:
                c["lane_spill"] += 1
            elif op.startswith("v_mad_u64_u32"):
                c["mad_u64_u32"] += 1
            elif op in ("v_rsq_f64_e32", "v_rcp_f64_e32", "v_sqrt_f64_e32"):
                c["rsq_f64"] += 1
            elif op.startswith("v_max_f64") or op.startswith("v_min_f64"):
                c["minmax_f64"] += 1
            elif "f64" in op:
                c["f64"] += 1
            else:
                c["int32"] += 1
        elif op.startswith("s_load"):
            c["smem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    ppl = 2
    res = {"kernel": m.group(1), "paths_per_lane": ppl, "loop_body_instructions": dict(c),
           "per_path_substep": {k: c[k] / ppl for k in ("mad_u64_u32", "int32", "f64", "rsq_f64", "minmax_f64", "lane_spill")}}
    res["per_path_substep"]["valu_total"] = sum(res["per_path_substep"].values())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
