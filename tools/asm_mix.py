"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only), to steer the instruction diet of the
issue-bound update kernels:   python tools/asm_mix.py /tmp/hx_fwdbwd.s [name filter ...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
filt = sys.argv[2:]
for m in re.finditer(r"^(_Z\w+|\w+):\s*; @\1\n(.*?)s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if filt and not any(k in name for k in filt):
        continue
    c = collections.Counter()
    for line in body.splitlines():
        t = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if not t:
            continue
        op = t.group(1)
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if "u64" in op or "b64" in op or "addc" in op:
                c["valu64"] += 1
            if op.startswith("v_cndmask"):
                c["cndmask"] += 1
            if op.startswith("v_mov") or op.startswith("v_accvgpr"):
                c["vmov"] += 1
            if "dpp" in line:
                c["dpp"] += 1
            if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
                c["readlane"] += 1
            if op.startswith("v_cmp"):
                c["vcmp"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_load") or op.startswith("s_memrealtime"):
            c["smem"] += 1
        elif op.startswith("s_nop"):
            c["nop"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
    keys = ["valu", "valu64", "cndmask", "vcmp", "vmov", "dpp", "readlane", "salu", "smem", "lds", "vmem", "mfma", "waitcnt", "nop"]
    print("%-70s " % name[:70] + " ".join("%s=%d" % (k, c[k]) for k in keys))
