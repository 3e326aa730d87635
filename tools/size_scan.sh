for wh in "960 540" "1920 1080" "2880 1620" "3840 2160"; do set -- $wh; echo -n "$1x$2 "; RT_W=$1 RT_H=$2 timeout -k 10 120 python tools/ab_time.py c3 8 pooled; done
