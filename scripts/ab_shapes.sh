for rep in 1 2; do for lib in base skip; do
MCBA_LIB=/root/repo/build_ab/libmcba_$lib.so MCBA_SHAPES="2,50,6,9;6,1000,6,9,1;6,1000,6,9;6,2130,5,7;6,10000,6,9;24,6250,10,20" timeout -k 10 200 python scripts/other_shapes.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', {k:(x['us_per_iteration'], x['kernels_us_by_hip_events']['k_solve_cam']) for k,x in d.items() if isinstance(x,dict)})
"
done; done
