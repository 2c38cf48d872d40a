for t0 in 6 8; do for t1 in 2 3; do for med in 1024 1280 1536; do for lg in 2048 2560; do
VIPRS_TEAM0=$t0 VIPRS_TEAM1=$t1 VIPRS_MEDIUM_BLOCK=$med VIPRS_LARGE_BLOCK=$lg timeout 100 python scratch/micro7.py 2>&1 | tail -1
done; done; done; done
