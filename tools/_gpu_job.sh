for lib in "" variants/lib_nocold_w3.so variants/lib_nocold_w4.so variants/lib_nocold_w5.so variants/lib_nocold_w6.so; do
  echo "== $lib"
  SOFTROD_HIP_LIB=${lib:+$PWD/$lib} python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'])"
done
