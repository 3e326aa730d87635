for rep in 1 2 3; do for v in 32 40 48 64; do
for w in c3 c2 c5 c4; do echo -n "job_min=$v "; RT_POOL_JOB_MIN=$v timeout -k 10 40 python tools/ab_time.py $w 10 pooled; done
echo -n "job_min=$v fused c3 "; RT_POOL_JOB_MIN=$v timeout -k 10 60 python tools/batch_time.py c3 8 5 | sed -n 2p
echo -n "job_min=$v N=8share "; RT_STRIP=9,8,0 RT_POOL_JOB_MIN=$v timeout -k 10 40 python tools/ab_time.py c3 10 pooled
done; done
