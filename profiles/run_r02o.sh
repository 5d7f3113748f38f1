cd "$GRAFT_REPO_ROOT"
bash profiles/collect.sh r02_triplane_R1_split triplane_R1_split "ngf::render_kernel" 2>&1 | tail -22
