#!/bin/bash
# Development tool (GPU box): what SQ_ACTIVE_INST_VALU counts -- on micro-benchmarks whose cost per instruction is known (tools/ubench2: v_mul 2.6
# cycles, v_min 4.4, v_rcp_f32 8.2, v_cndmask with vcc 22) and on the headline kernel.  Output: gpurun_out/valu_busy/{ubench,fused}.txt
OUT=gpurun_out/valu_busy
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/u -o u -- tools/build/ubench2 > $OUT/u.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/f -o f -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 5 --warmup 2 > $OUT/f.log 2>&1
python - <<'P'
import csv, glob, collections
for tag in ("u", "f"):
    ctr = glob.glob("gpurun_out/valu_busy/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not ctr: print(tag, "no counters"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(ctr[0])):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open("gpurun_out/valu_busy/%s.txt" % ("ubench" if tag == "u" else "fused"), "w") as o:
        for k, c in agg.items():
            m = {n: sum(v) / len(v) for n, v in c.items()}
            if m.get("SQ_INSTS_VALU", 0) < 1e5: continue
            o.write("%-46s INSTS_VALU %.3e  ACTIVE_INST_VALU/INSTS %.2f  ACTIVE_INST_ANY/INSTS %.2f  GUI_ACTIVE/8 %.3e  BUSY_CYCLES %.3e  WAVE_CYCLES %.3e  WAIT_INST_ANY %.3e  ACTIVE_VALU*4/1024/(GUI/8) %.2f\n" % (
                k, m["SQ_INSTS_VALU"], m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_INSTS_VALU"], m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_INSTS_VALU"], m.get("GRBM_GUI_ACTIVE", 0) / 8,
                m.get("SQ_BUSY_CYCLES", 0), m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_WAIT_INST_ANY", 0), m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / max(m.get("GRBM_GUI_ACTIVE", 1) / 8, 1)))
P
cat $OUT/fused.txt | head -3; grep -E "k_(mul|min|rcp_f32|cnd_e32|seq_real|fma|fma_vvv|mix_mul_min|seq_mmmn|dep_real)\(" $OUT/ubench.txt
