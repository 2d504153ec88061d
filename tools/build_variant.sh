# Dev helper: a variant build of the library for A/B timing on one box: tools/build_variant.sh NAME "-DRES_WAVES=5 ..."
# -> h264-fer_amd/var/libferhip_NAME.so (every object rebuilt with the flags; swap it in with
#    cp h264-fer_amd/var/libferhip_NAME.so h264-fer_amd/libferhip.so on the GPU box's scratch copy)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../h264-fer_amd/csrc"
rm -rf /tmp/fervar_$name && mkdir -p /tmp/fervar_$name ../var
for f in fer_api fer_refprep fer_me fer_resid fer_intra fer_cavlc fer_legacy fer_decode fer_fileio fer_mbunit; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -I../../include $flags -c $f.hip -o /tmp/fervar_$name/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../var/libferhip_$name.so /tmp/fervar_$name/*.o
ls -la ../var/libferhip_$name.so
