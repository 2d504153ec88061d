# Dev helper: a -DFER_PROBE build of the library next to the product one (h264-fer_amd/libferhip_probe.so).
# EVERY object is rebuilt: the kernels take FerDev by value, objects built against different fer_dev.h layouts fault.
set -e
cd "$(dirname "$0")/../h264-fer_amd/csrc"
rm -rf /tmp/ferprobe && mkdir -p /tmp/ferprobe
for f in fer_api fer_refprep fer_me fer_resid fer_intra fer_cavlc fer_legacy fer_decode fer_fileio fer_mbunit; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -I../../include -DFER_PROBE -c $f.hip -o /tmp/ferprobe/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libferhip_probe.so /tmp/ferprobe/*.o
ls -la ../libferhip_probe.so
