export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e2
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
PMC_SETS="$SQ1;$SQ2" CMD="python bench.py --workload hca_encode --streams 1000 --seconds 10 --no-cpu --no-verify --steps 3 --warmup 1" bash tools/prof_pmc.sh > gpurun_out/e2/pmc.log 2>&1
cp gpurun_out/pmc/pmc.json gpurun_out/e2/pmc.json
CRI_HIPCC_EXTRA=-DCRI_ENC_PROFILE python -m pycricodecs_amd.build --force > /dev/null 2>&1
python tools/debug/enc_phases.py 2 > gpurun_out/e2/phases2.txt 2>&1
python tools/debug/enc_phases.py 8 > gpurun_out/e2/phases8.txt 2>&1
python tools/debug/enc_phases.py 1 > gpurun_out/e2/phases1.txt 2>&1
cat gpurun_out/e2/phases2.txt gpurun_out/e2/phases8.txt gpurun_out/e2/phases1.txt
