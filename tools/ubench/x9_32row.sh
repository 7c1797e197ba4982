for v in 1073741824 4096; do echo "HX_ACT_X9_NRT2_ROWS=$v"; HX_ACT_X9_NRT2_ROWS=$v SIZES=4096 python tools/ubench/actp_time.py f32x9 2>&1 | tail -1; done
echo "f32 32-row:"; HX_ACT_F32_NRT2_ROWS=4096 SIZES=4096 python tools/ubench/actp_time.py f32 2>&1 | tail -1
HX_ACT_X9_NRT2_ROWS=4096 python -m pytest tests/test_x9_gpu.py -q 2>&1 | tail -3
