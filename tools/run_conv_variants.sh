for v in "" conv_NOPRE conv_NOSTORE conv_NOMMA; do
  if [ -z "$v" ]; then L=""; else L="I2T_LIB=$PWD/image2text_amd/csrc/libi2t_$v.so"; fi
  echo "== ${v:-base}"; env $L python tools/bench_conv.py 1024 2>&1 | grep -E "conv3 fwd  |conv3 bwd-data|conv3 fwd nhwc"
done
