export COSY_TUNE_LIB=1
python profiles/exp/det.py "COSY_WAVE_MASK=0x40 COSY_TAP_D=6 COSY_WAVE_DBG=16" "COSY_WAVE_MASK=0x40 COSY_TAP_D=6" 2>&1 | grep -v amdgpu | grep "B=256" | cut -c1-250
