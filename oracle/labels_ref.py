"""TEST INFRASTRUCTURE — label preparation oracle: ``TrainingModule.prepare_future_labels``
(streamingflow/trainer.py:283-394) as a function over the reference's OWN geometry helpers
(``streamingflow.utils.geometry``: warp_features / cumulative_warp_features[_reverse], imported from
/root/reference by oracle/gen_golden.py --only labels).  trainer.py itself needs pytorch_lightning (absent), so
only the dictionary plumbing of the method is restated here; the arithmetic is the reference's."""
import torch


def prepare_future_labels(G, batch, cfg, receptive_field, spatial_extent, encoder_downsample=8):
    labels = {}
    ego = batch["future_egomotion"]
    rf = receptive_field
    if cfg.LIFT.GT_DEPTH and "depths" in batch:
        d = batch["depths"][:, :rf, :, ::encoder_downsample, ::encoder_downsample]
        d = torch.clamp(d, cfg.LIFT.D_BOUND[0], cfg.LIFT.D_BOUND[1] - 1) - cfg.LIFT.D_BOUND[0]
        labels["depths"] = d.long().contiguous()

    def warp(x):
        past = G.cumulative_warp_features(x[:, :rf], ego[:, :rf], mode="nearest", spatial_extent=spatial_extent)
        fut = G.cumulative_warp_features_reverse(x[:, (rf - 1):], ego[:, (rf - 1):], mode="nearest", spatial_extent=spatial_extent)
        return past, fut
    p, f = warp(batch["segmentation"].float())
    labels["segmentation"] = torch.cat([p.long().contiguous()[:, :-1], f.long().contiguous()], dim=1)
    if cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED:
        p, f = warp(batch["pedestrian"].float())
        labels["pedestrian"] = torch.cat([p.long().contiguous()[:, :-1], f.long().contiguous()], dim=1)
    if cfg.INSTANCE_SEG.ENABLED:
        p, f = warp(batch["instance"].float().unsqueeze(2))
        labels["instance"] = torch.cat([p.long().contiguous()[:, :-1, 0], f.long().contiguous()[:, :, 0]], dim=1)
        for key in ("centerness", "offset"):
            p, f = warp(batch[key])
            labels[key] = torch.cat([p.contiguous()[:, :-1], f.contiguous()], dim=1)
    if cfg.INSTANCE_FLOW.ENABLED:
        p, f = warp(batch["flow"])
        labels["flow"] = torch.cat([p.contiguous()[:, :-1], f.contiguous()], dim=1)
    return labels
