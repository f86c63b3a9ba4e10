import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("config4_point","han_bf16","meta_edsr","b1_point","sparnet","inference","bf16x3"):
    v=d.get(k)
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,list,str))} if isinstance(v,dict) else v)
    if k=="b1_point" and isinstance(v,dict): print({a:round(b["value"],1) for a,b in v.items() if isinstance(b,dict) and "value" in b})
