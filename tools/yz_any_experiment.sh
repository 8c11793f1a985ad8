# A/B of the yz_lds_kernel variants of csrc/fft2p_any_yz.hip (MIL_FX_YZ_COLS4 / _PREF / _FUSE / _NT) under rocprofv3
B=${1:-"256,384,384;256,512,384;128,576,640;128,320,448;64,960,1024"}
for cfg in "1 0 0" "1 0 1" "1 1 1" "0 0 1" "0 0 0"; do
  set -- $cfg
  export MIL_FX_YZ_COLS4=$1 MIL_FX_YZ_PREF=$2 MIL_FX_YZ_FUSE=$3
  echo "== cols4=$1 pref=$2 fuse=$3"
  bash tools/prof_cmd.sh yzexp python tools/xy_any_probe.py "--boxes=$B" --iters=30 | grep "yz_lds\|it/s"
done
