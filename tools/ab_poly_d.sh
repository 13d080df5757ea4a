for lib in "" lib_w9 ""; do
  p=""; [ -n "$lib" ] && p=$PWD/rustradio_amd/$lib/librustradio_amd.so
  echo "== ${lib:-product}"
  RR_LIB_PATH=$p python tools/poly_probe.py 463,1000,2467,4000 9,10,11,12 2>/dev/null | awk '{print $1,$2,$3,$4,$5,$6}'
done
