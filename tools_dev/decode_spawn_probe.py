"""dev: ragged decode through SPAWNED DataLoader workers (no shared interpreter lock, no forked GPU runtime state) vs the thread pool"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from excel_amd.model import ExCEL_model
    from excel_amd.pipeline import TrainingFreePipeline
    from excel_amd.tools import synthetic
    from excel_amd.datasets.loader import ragged_batches, threaded_batches, DeviceFeeder
    dev = torch.device("cuda", 0)
    sd = synthetic.make_vit_state_dict(seed=0)
    model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=21, img_size=448, mode="train", device=dev, state_dict=sd,
                        text_features=synthetic.make_text_features(45))
    n = 256
    from excel_amd.datasets import voc
    with tempfile.TemporaryDirectory() as tmp:
        root, lists = os.path.join(tmp, "VOC2012"), os.path.join(tmp, "lists")
        synthetic.write_voc_tree(root, lists, n, seed=4321)
        ds = voc.VOC12SegDataset(root_dir=root, name_list_dir=lists, split="train", stage="val")
        pipe = TrainingFreePipeline(model, num_classes=21, smax=ds.max_k())
        order = [i for _ in range(4) for i in range(n)]
        for kind in ("threads16", "spawn8", "spawn12", "threads16"):
            if kind.startswith("threads"):
                batches = threaded_batches(ds, order, 32, num_threads=16)
            else:
                batches = ragged_batches(ds, order, 32, num_workers=int(kind[5:]), pin_memory=False, mp_context="spawn")
            torch.cuda.synchronize(); t0 = time.perf_counter(); t1 = None; cnt = 0
            for names, plan, images, cls_t, labels_t in DeviceFeeder(batches, dev):
                pipe.run_batch_ragged(images, plan, cls_t, labels_t)
                if t1 is None:
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                cnt += len(names)
            pipe.drain(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print(kind, "first batch %.2f s" % (t1 - t0), "steady %.1f img/s" % ((cnt - 32) / (t2 - t1)), "total %.1f img/s" % (cnt / (t2 - t0)), flush=True)
            # does an event wait still behave after the workers are gone?
            e = torch.cuda.Event(); e.record(); ta = time.perf_counter(); e.synchronize(); print("   event wait after: %.1f ms" % ((time.perf_counter() - ta) * 1e3), flush=True)


if __name__ == "__main__":
    main()
