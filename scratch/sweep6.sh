for t0 in 2 4 8 12; do VIPRS_TEAM0=$t0 timeout 200 python scratch/micro6.py 2>&1 | tail -4; done
for t1 in 2 3 6; do VIPRS_TEAM1=$t1 timeout 200 python scratch/micro6.py 2>&1 | tail -4; done
