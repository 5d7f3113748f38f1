cd "$GRAFT_REPO_ROOT"
bash profiles/collect.sh r02_uv_sphere_split uv_sphere_split "uv_render_kernel" 2>&1 | tail -24
