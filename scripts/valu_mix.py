"""Static instruction mix of the search kernel's hot variant (8-bit, DIA, sub-pel depth 4, unweighted): which share of its vector
instructions are of the kinds gfx950 issues in 2 cycles per wave64 (plain add/sub/and/or/xor/mov/ashr/cndmask in their VOP2/VOP1
encodings) and which take 4 (everything else: VOP3, packed math, DPP, SDWA, compares, min/max ...), as measured by
experiments/gen_valu_rate.py (table in experiments/README.md).  Writes profiles/search_valu_mix.json.
usage: python scripts/valu_mix.py      (needs hipcc, no GPU)"""
import collections
import json
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL_RATE = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_mov_b32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32",
             "v_ashrrev_i32_e32", "v_cndmask_b32_e32"}
src = """#include <hip/hip_runtime.h>
#include <stdint.h>
#include "x264hip.h"
#include "me_search.h"
template __global__ void me_rows_kernel<uint8_t, 0, 1, 0>( LaP, const SearchDesc<uint8_t> *, MeQueues, unsigned *, unsigned *, unsigned, unsigned long long * );
"""
with tempfile.TemporaryDirectory() as td:
    f = os.path.join(td, "k.hip")
    open(f, "w").write(src)
    asm = os.path.join(td, "k.s")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "x264_amd", "csrc"), "--cuda-device-only", "-S", f, "-o", asm],
                          stderr=subprocess.DEVNULL)
    text = open(asm).read()
body = text.split("s_endpgm")[0]
ops = collections.Counter(m.group(1) for m in re.finditer(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+)", body, re.M))
valu = {k: v for k, v in ops.items() if k.startswith("v_")}
n = sum(valu.values())
fast = sum(v for k, v in valu.items() if k in FULL_RATE)
out = {"kernel": "me_rows_kernel<uint8_t, HEX=0, MODE=1, WEIGHTED=0>", "static_instructions": sum(ops.values()), "static_valu": n,
       "static_valu_2_cycle": fast, "share_2_cycle": round(fast / n, 4), "cycles_per_valu": round(2 * fast / n + 4 * (1 - fast / n), 3),
       "s_nop": ops.get("s_nop", 0), "dpp": sum(v for k, v in valu.items() if k.endswith("_dpp")),
       "top": dict(collections.Counter(valu).most_common(24)),
       "note": "static mix as the proxy for the dynamic one; issue costs from experiments/gen_valu_rate.py (4 waves/SIMD: 1.0-1.1 ns per "
               "instruction for the 2-cycle kinds, 1.75-1.85 ns for the rest at ~2.3 GHz)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "search_valu_mix.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("static_valu", "share_2_cycle", "cycles_per_valu", "s_nop", "dpp")}))
