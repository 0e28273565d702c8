# the autoencoder's (cfg3) full-resolution few-tap layers, one by one
python tools/sep_conv_bench.py 8 1 3 1 1 1 1 1 1 0 0 160 192 160 4 20 fwd,dgrad,wgrad 2>/dev/null
python tools/sep_conv_bench.py 8 8 3 1 1 1 1 1 1 0 0 160 192 160 4 20 dgrad 2>/dev/null
python tools/sep_conv_bench.py 8 8 1 3 1 1 1 1 0 1 0 160 192 160 4 20 fwd,dgrad 2>/dev/null
python tools/sep_conv_bench.py 8 8 1 1 3 1 1 1 0 0 1 160 192 160 4 20 fwd,dgrad 2>/dev/null
python tools/sep_conv_bench.py 1 1 1 1 3 1 1 1 0 0 1 160 192 160 4 20 fwd,dgrad 2>/dev/null
python tools/sep_conv_bench.py 1 8 6 1 1 2 1 1 2 0 0 160 192 160 4 20 dgrad 2>/dev/null
python tools/sep_conv_bench.py 8 8 1 6 1 1 2 1 0 2 0 80 192 160 4 20 dgrad 2>/dev/null
python tools/sep_conv_bench.py 8 8 1 1 6 1 1 2 0 0 2 80 96 160 4 20 dgrad 2>/dev/null
