cd $GRAFT_REPO_ROOT
timeout -k 10 600 scripts/phase_timing.sh r06d --opt 21=0 > gpurun_out/r06d_phases.log 2>&1; grep -A8 '^PHASES' gpurun_out/r06d_phases.txt | grep 'k_tile' | cut -c1-1100
timeout -k 10 900 scripts/gpu_pmc.sh r06d "--opt 21=0" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" > gpurun_out/pmc_r06d.txt 2>&1; grep 'k_tile' gpurun_out/pmc_r06d.txt | cut -c1-600
python - <<'PY'
import json
d = json.load(open('gpurun_out/pmc_r06d/summary.json'))
for k in ('k_tile_rows', 'k_tile', 'k_tile_big', 'k_blur_small', 'k_blur_fused_dma<4>'):
    if k in d: print(k, d[k].get('counters'))
PY
