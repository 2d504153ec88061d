# Dev helper (GPU box): decode rate under several library variants.  usage: tools/ab_dec.sh "base noprio" "2 8"
cp h264-fer_amd/libferhip.so /tmp/libferhip_keep.so
for v in $1; do
  cp h264-fer_amd/var/libferhip_$v.so h264-fer_amd/libferhip.so
  echo "== $v"
  python tools/dec_rate.py $2 2>/dev/null | awk 'NR%2==0'
done
cp /tmp/libferhip_keep.so h264-fer_amd/libferhip.so
