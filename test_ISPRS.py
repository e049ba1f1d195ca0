"""Evaluation of a trained model on the ISPRS test tile - the reference's test_ISPRS.py on the
MI355X forward path: tile the normalised test image into non-overlapping patches, `model.predict(batch_size=1)`,
argmax over the segmentation output, accuracy / per-class F1 / recall / precision / confusion matrix against the
reference map, and the mosaic of the predictions.  Same flags, same function names, same numerics; the per-patch
matplotlib figures of the reference (test_ISPRS.py:302-400) are plotting and out of scope - the arrays
they would show are saved as .npy instead.

Reference: test_ISPRS.py:26-36 (Test), 39-45 (compute_metrics_hw), 48-87 (pred_recostruction), 89-99
(convert_preds2rgb), 102-151 (patch extraction), 174-190 (normalize_rgb), 193-210 (binarize_matrix), 215-300 (main).
"""
from __future__ import annotations

import argparse
import ast
import os
import sys

import numpy as np

# Dictionary used in training (test_ISPRS.py:262-263)
LABEL_DICT = {'(255, 255, 255)': 0, '(0, 255, 0)': 1, '(0, 255, 255)': 2, '(0, 0, 255)': 3, '(255, 255, 0)': 4}


def Test(model, patches, args):
    """test_ISPRS.py:26-36: dict of head outputs (multitask) or the argmax class map."""
    preds = model.predict(patches, batch_size=getattr(args, "batch_size", 1))
    if args.use_multitasking:
        print('Multitasking Enabled!')
        return preds
    print(preds.shape)
    predicted_class = np.argmax(preds, axis=-1)
    print(predicted_class.shape)
    return predicted_class


def compute_metrics_hw(true_labels, predicted_labels):
    """test_ISPRS.py:39-45 (= utils.compute_metrics): accuracy and per-class F1 / recall / precision in percent."""
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score
    accuracy = 100 * accuracy_score(true_labels, predicted_labels)
    f1score = 100 * f1_score(true_labels, predicted_labels, average=None)
    recall = 100 * recall_score(true_labels, predicted_labels, average=None)
    precision = 100 * precision_score(true_labels, predicted_labels, average=None)
    return accuracy, f1score, recall, precision


def pred_recostruction(patch_size, pred_labels, binary_img_test_ref, img_type=1):
    """test_ISPRS.py:48-87: row-major mosaic of the patches; the border that does not fill a patch stays zero."""
    height, width = binary_img_test_ref.shape
    nh, nw = height // patch_size, width // patch_size
    shape = (height, width) if img_type == 1 else (height, width, 3)
    img = np.zeros(shape)
    tiles = np.asarray(pred_labels)[:nh * nw].reshape((nh, nw, patch_size, patch_size) + shape[2:])
    mosaic = tiles.swapaxes(1, 2).reshape((nh * patch_size, nw * patch_size) + shape[2:])
    img[:nh * patch_size, :nw * patch_size] = mosaic
    print('Reconstruction Done!')
    return img


def convert_preds2rgb(img_reconstructed, label_dict):
    """test_ISPRS.py:89-99: class index -> the RGB triple of the label dictionary."""
    lut = np.zeros((max(label_dict.values()) + 1, 3), np.uint8)
    for key, value in label_dict.items():
        lut[value] = ast.literal_eval(key)
    print('Conversion to RGB Done!')
    return lut[img_reconstructed.astype(np.int64)]


def _tiles(img, patch_size):
    h, w = img.shape[:2]
    nh, nw = h // patch_size, w // patch_size
    t = img[:nh * patch_size, :nw * patch_size].reshape((nh, patch_size, nw, patch_size) + img.shape[2:])
    return t.swapaxes(1, 2).reshape((nh * nw, patch_size, patch_size) + img.shape[2:])


def extract_patches_test(binary_img_test_ref, patch_size):
    """test_ISPRS.py:102-125: non-overlapping reference patches, row-major, float64 like np.zeros there."""
    out = _tiles(binary_img_test_ref, patch_size).astype(np.float64)
    print(out.shape)
    return out


def extract_patches_train(img_test_normalized, patch_size):
    """test_ISPRS.py:128-151: non-overlapping image patches (N, ps, ps, C)."""
    out = _tiles(img_test_normalized, patch_size).astype(np.float64)
    print(out.shape)
    return out


def normalize_rgb(img, norm_type=1):
    """test_ISPRS.py:174-190.  norm_type 2 reproduces the reference's operator precedence (`img /= 127.5 - 1.`,
    i.e. a division by 126.5, not a map to [-1, 1]): a model trained with the reference's scripts saw exactly that."""
    if norm_type == 1:
        img /= 255.
    elif norm_type == 2:
        img /= 127.5 - 1.
    elif norm_type == 3:
        from sklearn.preprocessing import StandardScaler
        flat = img.reshape((img.shape[0] * img.shape[1]), img.shape[2])
        img = StandardScaler().fit_transform(flat).reshape(img.shape)
    return img


def binarize_matrix(img_train_ref, label_dict):
    """test_ISPRS.py:193-210: RGB reference map -> class indices (uint8); an unknown colour is a KeyError there too."""
    ref = np.asarray(img_train_ref)
    out = np.full(ref.shape[:2], 255, dtype=np.uint8)
    known = np.zeros(ref.shape[:2], bool)
    for key, value in label_dict.items():
        m = np.all(ref[..., :3] == np.array(ast.literal_eval(key)), axis=-1)
        out[m] = value
        known |= m
    if not known.all():
        i, j = np.argwhere(~known)[0]
        raise KeyError(str(tuple(int(v) for v in ref[i, j, :3])))
    return out


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--use_multitasking", help="Choose resunet-a model or not", action='store_true')
    parser.add_argument("--model_path", help="Model's filepath .h5", type=str, required=True)
    parser.add_argument("--dataset_path", help="Dataset directory path", type=str, required=True)
    parser.add_argument("-ps", "--patch_size", help="Size of Patches extracted from image and reference", type=int, default=256)
    parser.add_argument("--norm_type", choices=[1, 2, 3], type=int, default=1,
                        help="Types of normalization. Be sure to select the same type used in your training. "
                             "1 --> [0,1]; 2 --> [-1,1]; 3 --> StandardScaler() from scikit")
    parser.add_argument("--num_classes", help="Number of classes", type=int, default=5)
    parser.add_argument("--output_path", help="Path to where save predictions", type=str, default='results/preds_run')
    # additive
    parser.add_argument("--batch_size", type=int, default=1, help="patches per forward launch (reference: 1)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    from resunet_a_mltsk_keras_amd.keras_api import load_model
    from sklearn.metrics import confusion_matrix

    root_path = args.dataset_path
    img_test = np.load(os.path.join(root_path, 'Image_Test.npy')).astype(np.float32)          # (C, H, W) like load_npy_image
    if args.norm_type == 3:
        img_test_normalized = normalize_rgb(img_test.transpose((1, 2, 0)).copy(), norm_type=3).transpose((2, 0, 1))
    else:
        img_test_normalized = normalize_rgb(img_test, norm_type=args.norm_type)
    img_test_normalized = img_test_normalized.transpose((1, 2, 0))
    print(img_test_normalized.shape)
    img_test_ref = np.load(os.path.join(root_path, 'Reference_Test.npy')).transpose((1, 2, 0))
    print(img_test_ref.shape)
    binary_img_test_ref = binarize_matrix(img_test_ref, LABEL_DICT)

    patches_test = extract_patches_train(img_test_normalized, args.patch_size)
    patches_test_ref = extract_patches_test(binary_img_test_ref, args.patch_size)
    print(patches_test.shape)

    model = load_model(args.model_path, compile=False)
    model.summary()
    patches_pred = Test(model, patches_test, args)
    print('=' * 40)
    print('[TEST]')
    if args.use_multitasking:
        preds = patches_pred
        seg_pred = np.argmax(preds['seg'], axis=-1)
        print(f'seg shape argmax: {seg_pred.shape}')
    else:
        preds = None
        seg_pred = patches_pred

    true_labels = patches_test_ref.reshape(-1)
    predicted_labels = seg_pred.reshape(-1)
    metrics = compute_metrics_hw(true_labels, predicted_labels)
    cm = confusion_matrix(true_labels, predicted_labels)
    print('Confusion  matrix \n', cm)
    print()
    print('Accuracy: ', metrics[0])
    print('F1score: ', metrics[1])
    print('Recall: ', metrics[2])
    print('Precision: ', metrics[3])

    img_reconstructed = pred_recostruction(args.patch_size, seg_pred, binary_img_test_ref, img_type=1)
    img_reconstructed_rgb = convert_preds2rgb(img_reconstructed, LABEL_DICT)
    os.makedirs(args.output_path, exist_ok=True)
    h, w = img_reconstructed_rgb.shape[:2]
    with open(os.path.join(args.output_path, 'pred_seg_reconstructed.ppm'), 'wb') as f:      # dependency-free image
        f.write(f"P6\n{w} {h}\n255\n".encode() + img_reconstructed_rgb.tobytes())
    np.save(os.path.join(args.output_path, 'pred_seg_reconstructed.npy'), img_reconstructed.astype(np.uint8))
    np.save(os.path.join(args.output_path, 'confusion_matrix.npy'), cm)
    if preds is not None:                                      # what the reference's per-patch figures display
        for head in ('seg', 'bound', 'dist', 'color'):
            np.save(os.path.join(args.output_path, f'pred_{head}.npy'), preds[head].astype(np.float32))
    return {"accuracy": metrics[0], "f1": metrics[1], "recall": metrics[2], "precision": metrics[3], "confusion_matrix": cm}


if __name__ == "__main__":
    main(sys.argv[1:])
