#!/bin/bash
# ON THE GPU BOX: k_hca_parse variants by compile-time switch, on the bench material (tonal 10000 streams; sparse / mixed / noise 1000)
#   VARIANTS='|-DHCA_NO_PAIR|-DPARSE_WAVES=2' bash tools/debug/parse_ab.sh
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-verify --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['secondary']
print('$1', round(d['value']/1e6,1), d['roofline']['kernel_ms_per_step'], {k.replace('hca_decode_',''):(round(v['frames_per_s']/1e6,1), v.get('kernel_ms')) for k,v in s.items() if k.startswith('hca_decode_') and k.split('_')[-1] in ('spectra','mixed','noise')})"; }
IFS='|' read -ra VS <<< "${VARIANTS:-|-DHCA_NO_PAIR}"
for V in "${VS[@]}"; do
  CRI_HIPCC_EXTRA="$V" python -m pycricodecs_amd.build --force > /dev/null 2>&1
  run "[$V]"
done
python -m pycricodecs_amd.build --force > /dev/null 2>&1
