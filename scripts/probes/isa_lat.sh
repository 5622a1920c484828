#!/bin/bash
# instruction-order summary of conv3x3_lat_x3_kernel<8> (probe build): loads / waits / barriers / MFMAs in program order
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DLAT_PROBE "$@" -S --cuda-device-only lat_probe.hip -o /tmp/latp.s 2>/dev/null
awk '/lat_x3_kernelILi8EEEvNS_7LatArgsE:/,0' /tmp/latp.s | grep "global_load\|buffer_load\|s_barrier\|s_waitcnt\|s_cbranch\|v_mfma\|s_memrealtime\|ds_write\|ds_read\|global_store\|scratch_\|s_endpgm\|vgpr_count\|vgpr_spill" | awk '{print $1, ($1 ~ /waitcnt/ ? $2" "$3 : ""), ($1 ~ /vgpr/ ? $2 : "")}' | uniq -c
