for r in 1 2; do
for tp in 2 4; do for lib in product pb1; do
if [ $lib = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
OTH_TRUNK_TP=$tp python tools/netbench.py --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s|^|[$lib tp$tp r$r] |"
done; done; done
