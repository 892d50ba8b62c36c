# rocprofv3 --pmc passes over tools/pmc_gemm.py (SHAPE=qkv|proj); prints per-launch means of gemm_bt_pair_kernel
export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=/tmp/pmcg_$(echo $c | cut -c1-14 | tr ' ' '_'); rm -rf $d
  rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 tools/pmc_gemm.py > /tmp/pmcg.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_bt_pair_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"   {k:32s} launches={len(v)}  mean={sum(v)/len(v):.4g}")
PY
done
