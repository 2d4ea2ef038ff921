for p in 1 0 1 0; do echo "NB_BC_PRIO=$p"; NB_BC_PRIO=$p timeout -k 10 100 python tools/shard_times.py strict | grep -v amdgpu; done
echo 3D; for p in 1 0; do echo "NB_BC_PRIO=$p"; NB_FORCE_3D=1 NB_BC_PRIO=$p timeout -k 10 100 python tools/shard_times.py strict | grep -v amdgpu; done
