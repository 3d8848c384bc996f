timeout -k 10 200 python3 tools/prof_render.py bigmesh 2 && timeout -k 10 100 python3 tools/prof_render.py microfacet 4 && timeout -k 10 100 python3 tools/prof_render.py manylights 3
