import numpy as np
rng = np.random.default_rng(0)
proto = (rng.random((10, 784)) < 0.19) * rng.integers(100, 255, (10, 784))
lab = rng.integers(0, 10, 60000)
X = np.clip(proto[lab] + rng.normal(0, 40, (60000, 784)) * (rng.random((60000, 784)) < 0.3), 0, 255).astype(np.uint8)
np.save('/tmp/fake_mnist.npy', X); np.save('/tmp/fake_mnist_labels.npy', lab.astype(np.uint8))
