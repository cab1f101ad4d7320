
for cfg in "cvt_s1 64" "deit_small 128" "vil_tiny 64" "swin_base_w14 32" "swin_tiny_w14 128"; do
  set -- $cfg
  echo "== $1 B=$2"
  bash tools/ab_env.sh 2 ESVIT_WGRAD_STREAM=0 - -- --arch $1 --batch $2 --steps 15 --warmup 3 --no-roofline
done
