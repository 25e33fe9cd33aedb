"""Print the numbers of a stored final collection (profiles/<tag>_*) the documents quote.  python tools/final_summary.py [tag]"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06_final"
P = "profiles/" + tag + "_"
d = json.load(open(P + "bench_full.json"))
l = json.loads(open(P + "bench.json").read())
l20 = json.loads(open(P + "bench20.json").read())
r, h = d["roofline"], d["host_inclusive"]
b = h["batch_1000"]
one = d["one_batch_in_flight"]
print("srchash / box:", open(P + "box_info.txt").read().split("srchash:")[-1][:12] if "srchash" in open(P + "box_info.txt").read() else "?")
print(f"FA value {one['value']:.0f} ({one['ms_per_step']:.6f} ms)  3 handles {d['3_batches_in_flight']['value']:.0f}")
print(f"frac {r['frac']:.4f} events {r['events']['frac']:.4f} (family {r['events']['family_us_per_step']:.2f} us) mfma_util {r['mfma_util']:.3f} whole {r['whole_network_frac']:.4f} kernel_us {r['kernel_us_per_step']:.1f}")
print(f"host {h['value']:.0f} {h.get('passes')} at driver steps {h['at_driver_steps']['value']:.0f} {h['at_driver_steps'].get('passes')} frac of one {h['value'] / one['value']:.3f}")
print(f"B=1000: device {b['device_resident_one_in_flight']:.0f} ring {b['value']:.0f} {b.get('passes')} blocking {b['sync_call']['value']:.0f} loop {b['dropin_loop']['value']:.0f} {b['dropin_loop']['passes']}")
c = d["cpu_baseline"]
print(f"cpu {c['value']:.0f} ({c['cores']} cores) one thread x cores {c['per_core_x_cores']['value']:.0f}; ratios {d['speedup_vs_cpu_baseline']}")
print("reference on this GPU:", {k: v for k, v in d.items() if "reference" in k and isinstance(v, dict)}.keys(), l.get("reference_on_this_gpu"))
p = d["pileup"]
pr, ph = p["roofline"], p["host_inclusive"]
pb = ph["batch_1000"]
print(f"pileup one in flight {p['one_batch_in_flight']['value']:.0f} 3 handles {p['3_batches_in_flight']['value']:.0f} host {ph['value']:.0f} {ph.get('passes')} at driver steps {ph['at_driver_steps']['value']:.0f} frac {pr['frac']:.4f} whole {pr['whole_network_frac']:.4f} cpu {p.get('cpu_baseline', {}).get('value')}")
print(f"pileup B=1000: device {pb['device_resident_one_in_flight']:.0f} ring {pb['value']:.0f} blocking {pb['sync_call']['value']:.0f} loop {pb['dropin_loop']['value']:.0f} {pb['dropin_loop']['passes']}")
w = d["full_alignment_dwell"]
print(f"dwell {w['one_batch_in_flight']['value']:.0f} host {w.get('host_inclusive', {}).get('value')}")
print(f"bench20: {l20['value']} host {l20['host_inclusive']['value']} at driver steps {l20['host_inclusive']['at_driver_steps']} B1000 {l20['host_inclusive'].get('batch_1000')} pileup {l20['pileup']['value']} {l20['pileup'].get('host_inclusive')}")
for name, kk in (("FA", d["kernels"]), ("pileup", p["kernels"])):
    print(name, " ".join(f"{k.split('.')[1]} {v['avg_us']:.1f}/{(v['tflops'] or 0):.0f}/{(v['mfma_util'] or 0):.2f}" for k, v in kk.items()))
t, tp = json.load(open(P + "pmc_traffic.json")), json.load(open(P + "pmc_traffic_pileup.json"))
print(f"traffic {t['fabric_bytes_per_step'] / 1e6:.0f} MB L2 {t['l2_hit_rate']:.2f}; pileup {tp['fabric_bytes_per_step'] / 1e6:.0f} MB L2 {tp['l2_hit_rate']:.2f}")
wk = json.load(open(P + "worker_throughput_30_files.json"))
for k in ("full_alignment", "pileup"):
    print(k, {kk: (vv["loop_seconds"], vv["windows_per_s_in_the_loop"], vv["process_wall_seconds"], vv.get("torch_imported")) for kk, vv in wk[k].items() if isinstance(vv, dict)},
          wk[k]["vcf_identical"], wk[k]["vcf_records"], wk[k]["vcf_qual_last_digit_only"], len(wk[k]["vcf_call_differs"]))
print("240k:", wk["full_alignment_240k_windows_decoder_columns_only"]["libc3hip_decoder_columns"])
print(open(P + "pytest_gpu.txt").read().strip().splitlines()[-1])
print(open(P + "roofline_check.md").read().strip().splitlines()[-3:])
print(l["gt_concordance"], l["cpu_baseline"]["one_thread"], len(open(P + "bench.json").read()))
print(open("profiles/" + tag + "_batch_sweep.txt").read()[-2400:])
