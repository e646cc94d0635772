R=$PWD; T=$(mktemp -d); cd $T
python3 - <<PY
import sys; sys.path.insert(0,"$R")
from pixelbox_amd import weights, synth
open("w.pbxw","wb").write(weights.synthetic_blob(synth.SEED_WEIGHTS,128,128,256))
PY
g++ -O2 -std=c++17 -pthread -I $R/include $R/profiles/micro/ingest_staged.cpp -o ing -L $R/pixelbox_amd -lpixelbox_hip -Wl,-rpath,$R/pixelbox_amd -Wl,-rpath,/opt/rocm/lib 2>&1 | tail -2
for cfg in "8 1" "8 2" "12 1" "6 2"; do
  set -- $cfg
  for rep in 1 2; do ./ing w.pbxw 131072 256 256 $1 $2 1 | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('embedders $1 decoders $2:', round(r['images_per_s']/1e3,1),'k img/s')"; done
done
