# developer tool (one GPU call): the three rocprofv3 profile sets of tools/profile.sh + the default bench line
# usage: TAG=r02h bash tools/experiments/profile_all.sh   -> gpurun_out/${TAG}{,_bf16,_train}/, gpurun_out/${TAG}_bench_default.json
set -e
TAG=${TAG:-r02h}
TAG=$TAG bash tools/profile.sh
TAG=${TAG}_bf16 MODE="--dtype bf16" SUFFIX=_bf16 bash tools/profile.sh
TAG=${TAG}_train MODE="--mode train --no-train-compare" SUFFIX=_train bash tools/profile.sh
