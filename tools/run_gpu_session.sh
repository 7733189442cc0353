for rep in 1 2; do for c in 0 1 2 3; do echo "cfg $c"; CARE_LAT_CFG=$c python tools/latent_sweep.py 32768 84; done; done
