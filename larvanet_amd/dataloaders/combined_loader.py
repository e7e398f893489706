"""Threaded variant of the DIV2K sampler (dataloaders/combined_loader.py:122-163 of the
reference): N producer threads keep a bounded queue of ready batches per scale; the driver
pulls with get_queue_data().  Each producer owns its RandomState (the reference's threads share
the global numpy RNG and un-locked caches)."""
import queue
import threading

import numpy as np

from .div2k_train_loader import DIV2KLoader, augment_pair


def create_loader():
    return CombinedLoader()


class CombinedLoader(DIV2KLoader):
    def __init__(self):
        super().__init__()
        self.is_threaded = True
        self._threads = []
        self._stop = threading.Event()
        self._cache_lock = threading.Lock()

    def _add_args(self, parser):
        super()._add_args(parser)
        parser.add_argument("--data_num_queue_runners", type=int, default=6)
        parser.add_argument("--data_queue_size", type=int, default=16)

    def prepare(self, scales):
        super().prepare(scales)
        self._queues = {s: queue.Queue(maxsize=self.args.data_queue_size) for s in scales}

    def _image(self, key, path):
        with self._cache_lock:
            return super()._image(key, path)

    def _produce(self, scale, rng):
        while not self._stop.is_set():
            lrs, hrs = [], []
            for _ in range(self._batch):
                lr, hr, _ = self.get_image_pair(rng.randint(self.get_num_images()), scale)
                a, b = augment_pair(rng, lr, hr, scale, self._patch)
                lrs.append(a)
                hrs.append(b)
            while not self._stop.is_set():
                try:
                    self._queues[scale].put((lrs, hrs), timeout=0.2)
                    break
                except queue.Full:
                    continue

    def start_training_queue_runner(self, batch_size, input_patch_size):
        self.stop_queue_runners()
        self._stop.clear()
        self._batch, self._patch = batch_size, input_patch_size
        for scale in self.scale_list:
            for i in range(self.args.data_num_queue_runners):
                rng = np.random.RandomState(self.rng.randint(2 ** 31 - 1))
                t = threading.Thread(target=self._produce, args=(scale, rng), daemon=True)
                t.start()
                self._threads.append(t)

    def stop_queue_runners(self):
        self._stop.set()
        for t in self._threads:
            t.join()
        self._threads = []

    def get_queue_data(self, scale):
        if not self._threads:
            return None
        return self._queues[scale].get()
