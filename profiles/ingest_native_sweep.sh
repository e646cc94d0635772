set -e
cd $GRAFT_REPO_ROOT
python3 -c "
from pixelbox_amd import weights as W, synth
open('/tmp/w.pbxw','wb').write(W.synthetic_blob(synth.SEED_WEIGHTS,128,128,256))"
g++ -O2 -std=c++17 -pthread -I include profiles/micro/ingest_staged.cpp -o /tmp/ingest_staged -L pixelbox_amd -lpixelbox_hip -Wl,-rpath,$PWD/pixelbox_amd -Wl,-rpath,/opt/rocm/lib
for cfg in "16384 2 1" "65536 2 1" "65536 3 1" "65536 2 0" "65536 4 1"; do set -- $cfg; echo "640x480 n=$1 dec=$2 mode=$3: $(timeout 200 /tmp/ingest_staged /tmp/w.pbxw $1 640 480 8 $2 $3 2>&1 | tail -1)"; done
for cfg in "131072 1 1" "131072 2 1" "131072 1 0"; do set -- $cfg; echo "256x256 n=$1 dec=$2 mode=$3: $(timeout 200 /tmp/ingest_staged /tmp/w.pbxw $1 256 256 8 $2 $3 2>&1 | tail -1)"; done
