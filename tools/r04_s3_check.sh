set -u
O=gpurun_out/r04_s3b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "fp16x3 or split_precision or exact or gemm" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in "" real-time-deepfake-speech-detection_amd/lib/libafx_s1.so; do
  echo "== AFX_LIB=$L" >> $O/scale_ab.txt
  AFX_LIB=$L timeout -k 10 600 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_models.py tests/test_gpu_kernels.py -m gpu -q -s --timeout 600 -k "unconditional_parity_modes and fp16x3 or 2dp and fp16x3 or split_precision" 2>&1 | grep -E "config 3|EER|split precision|passed|failed|Error" >> $O/scale_ab.txt
done
cat $O/scale_ab.txt
timeout -k 10 400 python tools/diag_s3_knobs.py > $O/s3_knobs.txt 2>&1; cat $O/s3_knobs.txt
