cd ${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do for w in 256 64; do echo "WG=$w $(EVOGP_TC_PACKED_WG=$w python scripts/pool_sweep.py 2>&1 | grep trees | tr '\n' ';' | cut -c1-700)"; done; done
