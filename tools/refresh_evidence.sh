#!/bin/bash
# usage: refresh_evidence.sh TAG : everything profiles/ holds for a round, measured from the current tree in one go --
# PMC HBM traffic (training step, K3, A-NeRF frame), the GPU test-suite's measured parity values, the bench line of every config,
# rocprofv3 kernel stats of the render frame and of the training step.  Outputs under gpurun_out/evidence_TAG/.
TAG=$1
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/evidence_$TAG
mkdir -p $E
cd $R
bash tools/pmc_train.sh $TAG > $E/pmc_train.log 2>&1 && cp gpurun_out/pmc_train_$TAG.json profiles/${TAG}_pmc_train.json && cp gpurun_out/pmc_train_$TAG.json $E/${TAG}_pmc_train.json
bash tools/pmc_hbm.sh $TAG > $E/pmc_hbm.log 2>&1 && cp gpurun_out/pmc_hbm_$TAG.json profiles/${TAG}_pmc_hbm.json && cp gpurun_out/pmc_hbm_$TAG.json $E/${TAG}_pmc_hbm.json
bash tools/pmc_anerf.sh $TAG > $E/pmc_anerf.log 2>&1 && cp gpurun_out/pmc_anerf_$TAG.json profiles/${TAG}_pmc_anerf.json && cp gpurun_out/pmc_anerf_$TAG.json $E/${TAG}_pmc_anerf.json
python -m pytest tests -m gpu -q -s 2>&1 | grep -E "raw_err|vs oracle|deviation|worst|passed|failed" > $E/${TAG}_parity_measured.txt
for c in 1 2 3 4 5; do
  python bench.py --config $c > $E/bench$c.log 2>&1
  tail -1 $E/bench$c.log | python -c "import sys, json; print(json.dumps(json.loads(sys.stdin.read()), indent=1))" > $E/${TAG}_bench_config$c.json
done
python bench.py --config 4 --gpus 2 --scaling strong --debug-single-device > $E/bench4s.log 2>&1
tail -1 $E/bench4s.log | python -c "import sys, json; print(json.dumps(json.loads(sys.stdin.read()), indent=1))" > $E/${TAG}_bench_config4_strong_2ranks_1gpu_debug.json
bash tools/prof.sh ev_$TAG > $E/prof_frame.log 2>&1; cp gpurun_out/prof_ev_$TAG/ev_${TAG}_kernel_stats.csv $E/${TAG}_kernel_stats.csv
bash tools/prof_train.sh ev_$TAG > $E/prof_train.log 2>&1; cp gpurun_out/prof_ev_$TAG/ev_${TAG}_kernel_stats.csv $E/${TAG}_train_kernel_stats.csv
bash tools/prof_config.sh ev3_$TAG 3 > $E/prof_config3.log 2>&1; cp gpurun_out/prof_ev3_$TAG/ev3_${TAG}_kernel_stats.csv $E/${TAG}_config3_kernel_stats.csv
bash tools/timeline_train.sh ev_$TAG > /dev/null 2>&1; cp gpurun_out/tl_ev_$TAG.txt $E/${TAG}_train_timeline.txt
bash tools/prof_anerf.sh eva_$TAG > $E/prof_anerf.log 2>&1; cp gpurun_out/prof_eva_$TAG/eva_${TAG}_kernel_stats.csv $E/${TAG}_anerf_kernel_stats.csv
bash tools/pmc_gather.sh $TAG > $E/pmc_gather.log 2>&1; cp gpurun_out/gather_hbm_$TAG.json $E/${TAG}_gather_hbm.json
PMC_FRAME=1 bash tools/pmc_kernel.sh k2_$TAG k_assign16 > $E/${TAG}_pmc_sq_k2_frame.txt 2>&1
bash tools/pmc_kernel.sh k2b_$TAG k_assign_bwd > $E/${TAG}_pmc_sq_k2_bwd.txt 2>&1
bash tools/pmc_mlp32.sh $TAG > $E/${TAG}_pmc_sq_k3_forms.txt 2>&1
(for w in mlp32 mlp16 mlp32z mlp16z idle; do python tools/clock_trace.py $w 4; done) > $E/${TAG}_k3_forms_clock_power.txt 2>&1
(cd tools/probe && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value -o cview_probe cview_probe.hip 2>/dev/null; timeout 600 ./cview_probe 2000) > $E/${TAG}_pk_f32_erratum.txt 2>&1
# gpurun merges at most 64 MiB back: the raw rocprofv3 directories stay on the box, the summaries above are what is kept
find gpurun_out -mindepth 1 -maxdepth 1 -type d ! -name "evidence_$TAG" -exec rm -rf {} +
find gpurun_out -maxdepth 1 -type f -size +4M -delete
du -sh gpurun_out | tail -1
grep -h '"ms_per_step"' $E/*_bench_config*.json
tail -3 $E/${TAG}_parity_measured.txt
