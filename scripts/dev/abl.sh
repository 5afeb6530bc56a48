for a in 0 1 2 4 6 7; do echo -n "split2 ring ABL=$a "; SFM_KNN_ABL=$a python scripts/run_knn_steps.py 50 10000 10000 uniform | cut -c1-80; done
