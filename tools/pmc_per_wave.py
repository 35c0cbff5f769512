import json,sys
d=json.load(open(sys.argv[1]))["kernels"]
for k,v in d.items():
    w=v["SQ_WAVES"]
    print(k, {c: round(x/w,1) for c,x in v.items() if c.startswith("SQ_")}, "hbmGB", round(v["hbm_bytes_per_launch"]/1e9,3), "hit", round(v["TCC_HIT_sum"]/(v["TCC_HIT_sum"]+v["TCC_MISS_sum"]),3), "gui", v["GRBM_GUI_ACTIVE"]/8)
