for ug in 512 1024 2048; do echo "UGRID $ug: $(DAISY_STAGED_UGRID=$ug python tools/probe_staged.py c2 2>&1 | grep staged/indexed)"; done
for ig in 1024 2048 4096 65536; do echo "IGRID $ig: $(DAISY_STAGED_IGRID=$ig python tools/probe_staged.py c2 2>&1 | grep staged/indexed)"; done
