"""The legs of bench.py, one module per workload (round 6: bench.py was 1 660 lines): common (constants, barrier / reduction helpers, the
CPU baseline of the headline), knn (BASELINE configs[1]: the headline), geometry (triangulation, dense BA sweep: configs[3]), scale
(configs[4], all-pairs, the dry run of the N-rank protocol), features (SIFT, the 57-camera driver from features and from pixels:
configs[2]), line (the compact stdout line + the full record)."""
