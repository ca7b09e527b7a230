"""Same-sample-stream image parity of the full-size C3 / C4 crops (GPU vs oracle), with the 20 worst pixels explained
sample by sample: gpurun_out/parity_r3.json (committed as profiles/parity_r3.json).
A pixel's samples are compared one by one (box filter, one sample per render, sample_first = k: the film then holds the
radiance of sample k of every pixel).  A sample 'flips' when the two renderers disagree by more than 1e-3 relative: a branch
decided by the last bit (a hit across an edge, the lobe a BSDF sample picks, a Newton step of the visible-normal sampler),
after which the sample is simply a different, equally valid one; the rest differs by rounding only ('systematic')."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ray-tracer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, kkdirect, kkoracle, helpers
from kkdirect import scenes, abi

EPS = 1e-3


def analyse(name, sc, crop, spp):
    s = sc.sensor
    s.crop_offset_x, s.crop_offset_y, s.crop_width, s.crop_height = crop
    o = kkoracle.Oracle(sc)
    ref, _ = o.render(spp, seed=1)
    with kkdirect.Context(device=0, seed=1).upload(sc) as ctx:
        film, _ = ctx.render(spp)
    a, b = helpers.develop(film), helpers.develop(ref)
    err = ((a - b) ** 2 / (b ** 2 + EPS)).sum(-1) / 3.0               # per pixel; its mean is the relMSE
    relmse = float(err.mean())
    # per-sample radiance of every pixel of the crop, both renderers
    s.rfilter_type = abi.RFILTER_BOX; s.rfilter_radius = 0.5
    o2 = kkoracle.Oracle(sc)
    ga, gb = [], []
    with kkdirect.Context(device=0, seed=1).upload(sc) as ctx:
        for k in range(spp):
            ga.append(ctx.render(1, sample_first=k, sample_count_total=spp)[0][..., :3].copy())
            gb.append(o2.render(1, seed=1, sample_first=k, sample_count_total=spp)[0][..., :3].copy())
    ga, gb = np.stack(ga, 0).astype(np.float64), np.stack(gb, 0).astype(np.float64)      # spp x H x W x 3
    rel = np.abs(ga - gb).max(-1) / (np.abs(gb).max(-1) + 1e-2)
    flipped = rel > 1e-3
    rest_rel = np.where(flipped, 0.0, rel)
    # what the image error would be without the flipped samples (oracle's value in their place)
    gmix = np.where(flipped[..., None], gb, ga)
    out = {"scene": name, "crop": list(crop), "spp": spp, "relmse_same_stream": relmse,
           "samples": int(flipped.size), "flipped_samples": int(flipped.sum()), "flipped_share": float(flipped.mean()),
           "relmse_box_filter_all_samples": helpers.rel_mse(ga.mean(0), gb.mean(0)),
           "relmse_box_filter_without_flipped_samples": helpers.rel_mse(gmix.mean(0), gb.mean(0)),
           "max_relative_difference_of_unflipped_samples": float(rest_rel.max()),
           "median_relative_difference_of_unflipped_samples": float(np.median(rel[~flipped])), "worst_pixels": []}
    order = np.argsort(err.ravel())[::-1][:20]
    counts = {}
    for idx in order:
        y, x = divmod(int(idx), err.shape[1])
        # the Gaussian filter spreads a sample over 5 x 5 pixels: look at the samples of the pixel and its neighbours
        y0, y1, x0, x1 = max(0, y - 2), min(err.shape[0], y + 3), max(0, x - 2), min(err.shape[1], x + 3)
        nf = int(flipped[:, y0:y1, x0:x1].sum())
        big = float(rel[:, y0:y1, x0:x1].max())
        cls = "sample flip (discontinuity decided by the last bit)" if nf > 0 else "systematic (rounding only)"
        counts[cls] = counts.get(cls, 0) + 1
        out["worst_pixels"].append({"x": x + crop[0], "y": y + crop[1], "share_of_relmse": float(err[y, x] / err.sum()),
                                    "flipped_samples_in_footprint": nf, "samples_in_footprint": int(flipped[:, y0:y1, x0:x1].size),
                                    "largest_relative_sample_difference": big, "class": cls})
    out["worst_pixel_classes"] = counts
    out["share_of_relmse_in_20_worst_pixels"] = float(err.ravel()[order].sum() / err.sum())
    s.rfilter_type = abi.RFILTER_GAUSSIAN; s.rfilter_radius = 2.0
    s.crop_offset_x, s.crop_offset_y, s.crop_width, s.crop_height = 0, 0, s.film_width, s.film_height
    return out


if __name__ == "__main__":
    res = []
    t = time.time()
    res.append(analyse("C3 full size: C2 geometry + 10 000 single-triangle emitters, 1920x1080", scenes.many_lights(1920, 1080, interior_tris=260_000), (832, 476, 256, 128), 16))
    print(json.dumps({k: v for k, v in res[-1].items() if k != "worst_pixels"}), "%.0f s" % (time.time() - t), flush=True)
    res.append(analyse("C4 full size: 10.7 M triangles, 4096x2048 environment map, 3840x2160", scenes.scanned_mesh(3840, 2160), (1792, 1016, 256, 128), 16))
    print(json.dumps({k: v for k, v in res[-1].items() if k != "worst_pixels"}), "%.0f s" % (time.time() - t), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"tool": "tests/tools/parity_r3.py", "results": res}, open(os.path.join(ROOT, "gpurun_out", "parity_r3.json"), "w"), indent=1)
