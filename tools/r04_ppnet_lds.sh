#!/bin/bash
# LDS bank-conflict cycles against LDS-array cycles of every kernel of a PPNet batch
ROOT=$PWD; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/r04; cd /tmp; rm -rf /tmp/ppl
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/ppl -- python3 $ROOT/tools/profile_ppnet.py 256 > /dev/null 2>&1 || { echo failed; exit 1; }
python3 $ROOT/tools/pmc_avg.py /tmp/ppl 2>&1 | grep -v "at::native\|rocclr\|std::array" > $ROOT/gpurun_out/r04/ppnet_lds.txt
python3 - <<'PY'
import re,ast
for l in open("/root/repo/gpurun_out/r04/ppnet_lds.txt"):
    m=re.match(r"(.*?) (\{.*\}) n=(\d+)", l.strip())
    if not m: continue
    d=ast.literal_eval(m.group(2))
    if d.get("SQ_LDS_IDX_ACTIVE",0)>0:
        print("%-44s conflicts %12d  active %12d  frac %.2f  n=%s" % (m.group(1)[-44:], d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"], d["SQ_LDS_BANK_CONFLICT"]/d["SQ_LDS_IDX_ACTIVE"], m.group(3)))
PY
