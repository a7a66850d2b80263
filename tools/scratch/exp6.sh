bash tools/scratch/exp5.sh 2>&1 | grep "triplane\|whole\|checksum"
FVP_NO_FINE_CACHE=1 bash tools/scratch/exp5.sh 2>&1 | grep "triplane\|checksum"
