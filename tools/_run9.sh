python tools/aten_trace.py --dtype bf16 2>&1 | grep -v amdgpu.ids | head -50
echo ======; python tools/aten_trace.py --dtype mixed 2>&1 | grep -v amdgpu.ids | head -40
