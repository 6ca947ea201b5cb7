cd $GRAFT_REPO_ROOT
bash scratch/ab_rh.sh "scf_base scf_xcd"
bash scratch/ab.sh "scf_base scf_xcd scf_base scf_xcd" | grep -v "nodes "
