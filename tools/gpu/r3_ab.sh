cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"; SDX_LIB_PATH=$PWD/seqdex_amd/lib/libseqdex_$v.so timeout 100 python tools/time_physics.py 1024 8 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_cycles_env0_substep0']; print(d['k_physics_ms'], {k:p[k] for k in ('collide','solve','solve_setup','it_AC','it_D','it_robot','it_total','narrow','mass_matrix') if k in p})"
done
