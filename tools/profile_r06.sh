#!/bin/bash
# Round-6 profiles (GPU box): kernel inventory of the step, then PMC passes -- one counter set per pass, as the guide
# prescribes -- over the dominant conv kernels (`bench.py --roofline-only`) and the weight-gradient launch pair
# (`bench.py --wgrad-only`).  Summaries land in gpurun_out/r06_prof/ (copy the ones to keep into profiles/).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_prof
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/step.log 2>&1 || { echo step trace failed; tail -5 $O/step.log; exit 1; }
cp $O/step/*/*kernel_stats.csv $O/bench_kernel_stats.csv
(cd $R && python tools/step_timeline.py gpurun_out/r06_prof/step > gpurun_out/r06_prof/bench_timeline_summary.txt 2>&1)
echo "step trace done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/conv$i -- python3 $R/bench.py --roofline-only > $O/conv$i.log 2>&1 || { echo conv pass $i failed; tail -5 $O/conv$i.log; exit 1; }
  echo "conv pass $i done"
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/wgrad$i -- python3 $R/bench.py --wgrad-only > $O/wgrad$i.log 2>&1 || { echo wgrad pass $i failed; tail -5 $O/wgrad$i.log; exit 1; }
  echo "wgrad pass $i done"
done
cd $R
python tools/pmc_summary.py gpurun_out/r06_prof/pmc_conv.csv fetch=$O/conv1 write=$O/conv2 mfma=$O/conv3 lds=$O/conv4
python tools/pmc_summary.py gpurun_out/r06_prof/pmc_wgrad.csv fetch=$O/wgrad1 write=$O/wgrad2 mfma=$O/wgrad3
# un-profiled in-kernel stamp profiles (tools/_diag/liblarva_step.so = -DLARVA_DIAG=544, built in the build container)
python tools/diag_overlap.py gpurun_out/r06_prof/dual_chain_overlap.txt > /dev/null 2>&1 || echo "diag_overlap failed"
python tools/diag_step.py gpurun_out/r06_prof/step_timeline_stamped.txt > /dev/null 2>&1 || echo "diag_step failed"
python tools/step_marks.py gpurun_out/r06_prof/step_marks_product.txt > /dev/null 2>&1 || echo "step_marks failed"
# the raw traces are large: keep the summaries only
rm -rf $O/step $O/conv1 $O/conv2 $O/conv3 $O/conv4 $O/wgrad1 $O/wgrad2 $O/wgrad3
ls -la $O
# Round 6: the inference path (VERDICT r4 missing 3): kernel inventory of 33 full-image forwards of V1 and V2 (single
# stream: profiles faithfully) + PMC passes over the persistent whole-tensor conv launch
cd /tmp
for m in LarvaNet LarvaNetV2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer_$m -- python3 $R/tools/infer_full_image.py $m > $O/infer_$m.log 2>&1 || { echo infer trace $m failed; tail -5 $O/infer_$m.log; exit 1; }
  cp $O/infer_$m/*/*kernel_stats.csv $O/infer_${m}_kernel_stats.csv
  echo "infer trace $m done: $(tail -1 $O/infer_$m.log)"
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/infer_pmc$i -- python3 $R/tools/infer_full_image.py LarvaNet > $O/infer_pmc$i.log 2>&1 || { echo infer pmc pass $i failed; tail -5 $O/infer_pmc$i.log; exit 1; }
  echo "infer pmc pass $i done"
done
cd $R
python tools/pmc_summary.py gpurun_out/r06_prof/pmc_infer.csv fetch=$O/infer_pmc1 write=$O/infer_pmc2 mfma=$O/infer_pmc3
rm -rf $O/infer_LarvaNet $O/infer_LarvaNetV2 $O/infer_pmc1 $O/infer_pmc2 $O/infer_pmc3
# one stamped full-image layer per epilogue (tools/_diag/libconv_diag32.so = -DLARVA_DIAG=32, built in the build container)
if [ -f tools/_diag/libconv_diag32.so ]; then
  for e in relu res1 res2; do python tools/diag_wide.py $e >> $O/infer_wide_layer_stamps.txt 2>&1 || echo "diag_wide $e failed"; done
fi
python tools/bench_wide_layer.py > $O/ab_persist.txt 2>&1 || echo "bench_wide_layer failed"
ls -la $O
