# rocprofv3 --pmc passes over tools/pmc_attn.py; prints per-launch means of the S = 257 attention kernel
export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  d=/tmp/pmca_$(echo $c | cut -c1-14 | tr ' ' '_'); rm -rf $d
  rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 tools/pmc_attn.py > /tmp/pmca.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "attention_s257" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"   {k:32s} launches={len(v)}  mean={sum(v)/len(v):.4g}")
PY
done
