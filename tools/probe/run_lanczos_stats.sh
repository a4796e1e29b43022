cd $GRAFT_REPO_ROOT
python3 tools/probe/lanczos_stats.py 64 2>&1 | grep -v amdgpu.ids
