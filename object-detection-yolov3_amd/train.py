#!/usr/bin/env python3
"""train.py -- YOLOv3 training loop on MI355X.  Reference: train.py:28-267 (same flags and semantics).

    python train.py --train_database D/train-x.lmdb --test_database D/test-x.lmdb --output_dir OUT
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...   # 8 GPUs

Kept from the reference: anchors [(64,384),(384,64)] (train.py:33), READER_COUNT = 3 reader processes per GPU (:16),
global batch = batch_size x replicas (:41), Adam warm-up at lr/10 for min(1000, N) steps of epoch 0 (:107-113), the
"step > N: break" loop bound (N+1 steps, Q16), NaN-loss abort (:124-125), per-step metric print, test loop of
image_count/batch_size(+1) batches (:76,:144), test_loss.csv (:170-173), best checkpoint on a new minimum (:178-182),
early stopping with CONVERGENCE_TOLERANCE 1e-4 (:185-197), final export of the best checkpoint (:208-221).
Changed: MirroredStrategy -> one process per GPU + RCCL (yolo3.parallel); TF checkpoint / SavedModel -> .npz weight
files (<out>/checkpoint/ckpt.npz, <out>/saved_model/yolov3.npz); TensorBoard event files -> <out>/scalars-<ts>/{train,test}.csv.
"""
import argparse
import datetime
import os
import time

import numpy as np

READER_COUNT = 3  # per gpu (train.py:16)
CONVERGENCE_TOLERANCE = 1e-4  # train.py:185


def best_epoch_of(test_loss, tolerance=CONVERGENCE_TOLERANCE):
    """train.py:185-192: the FIRST epoch whose test loss is within `tolerance` of the minimum."""
    error_from_best = np.abs(np.asarray(test_loss) - np.min(test_loss))
    error_from_best[error_from_best < tolerance] = 0
    return int(np.where(error_from_best == 0)[0][0])


def should_stop(test_loss, early_stopping_count, tolerance=CONVERGENCE_TOLERANCE):
    """train.py:193-197: stop once more than `early_stopping_count` epochs have passed since the best one."""
    return len(test_loss) - best_epoch_of(test_loss, tolerance) > early_stopping_count


def is_new_minimum(test_loss):
    """train.py:178: checkpoint when the newest test loss is the (first) minimum of the series."""
    return (len(test_loss) - 1) == int(np.argmin(test_loss))


def abort_on_nan(loss_value, message):
    """train.py:124-125,151-152."""
    if np.isnan(float(loss_value)):
        raise RuntimeError(message)


def effective_reader_count(requested, cpus, local_world):
    """Reader processes per GPU.  The reference starts READER_COUNT = 3 per GPU (train.py:16) for a TensorFlow step; a ~17 ms
    step with augmentation on (~20 ms of CPU per image) wants about 12 (tools/train_throughput.py).  Two readers (train, test)
    run per rank and every rank of the node shares the host, so both the default and an explicit --reader_count are capped by
    the cores one rank may use: cpus // local_world minus one for the rank's own main + prefetch threads, never below 1."""
    share = max(1, (cpus or 1) // max(local_world, 1) - 1)
    if requested is None:
        requested = max(READER_COUNT, min(12, share - 1))
    return max(1, min(int(requested), share))


def train_model(batch_size, test_every_n_steps, train_database_filepath, test_database_filepath, output_folder, early_stopping_count,
                learning_rate, use_augmentation, max_epochs=None, reader_count=None, backend='nccl'):
    os.makedirs(output_folder, exist_ok=True)
    anchors = [(64, 384), (384, 64)]

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    global_batch_size = batch_size * world
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    asked = reader_count
    reader_count = effective_reader_count(reader_count, os.cpu_count() or 8, local_world)
    if asked is not None and reader_count != asked:
        print('reader_count {} capped to {} ({} cpus / {} ranks on this node)'.format(asked, reader_count, os.cpu_count(), local_world))

    # readers first: their worker processes are forked before this process touches the GPU
    from yolo3 import imagereader
    print('Setting up test image reader')
    # one reader per rank: the unshuffled test reader takes this rank's stride of the key list (the reference splits one
    # global test batch over its replicas, train.py:64-66)
    test_reader = imagereader.ImageReader(test_database_filepath, anchors, use_augmentation=False, shuffle=False, num_workers=reader_count,
                                          num_shards=world, shard_index=rank)
    print('Test Reader has {} images'.format(test_reader.get_image_count()))
    print('Setting up training image reader')
    train_reader = imagereader.ImageReader(train_database_filepath, anchors, use_augmentation=use_augmentation, shuffle=True,
                                           num_workers=reader_count, balance_classes=True)
    print('Train Reader has {} images'.format(train_reader.get_image_count()))
    training_checkpoint_filepath = None
    try:
        print('Starting Readers')
        train_reader.startup()
        test_reader.startup()

        import torch
        import torch.distributed as dist
        from yolo3 import model
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        from yolo3 import streams
        streams.reserve()      # side / comm streams bind to hardware queues before a communicator creates its own (yolo3/streams.py)
        strategy = None
        if world > 1:
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            if backend == 'nccl':                  # = RCCL over xGMI, one GPU per rank
                # a launcher that hands every rank its OWN device through *_VISIBLE_DEVICES shows exactly one device per rank: fine.
                # A mask that shows several devices but fewer than the ranks would put two nccl ranks on one device.
                isolated = any(os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'))
                ndev = torch.cuda.device_count()
                if ndev < int(os.environ.get('LOCAL_WORLD_SIZE', str(world))) and not (isolated and ndev == 1):
                    raise RuntimeError('backend nccl (RCCL) needs one GPU per rank: %d visible for %s ranks; use --backend gloo to rehearse on fewer'
                                       % (ndev, os.environ.get('LOCAL_WORLD_SIZE', str(world))))
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank % torch.cuda.device_count()))
            else:
                dist.init_process_group(backend)
            from yolo3.parallel import DataParallel
            strategy = DataParallel()

        train_dataset = train_reader.get_tf_dataset().batch(batch_size).prefetch(reader_count)
        test_dataset = test_reader.get_tf_dataset().batch(batch_size).prefetch(reader_count)

        print('Creating model')
        number_classes = train_reader.get_number_classes()
        yolo = model.YoloV3(global_batch_size, train_reader.get_image_size(), number_classes, anchors, learning_rate)
        if strategy is not None:
            strategy.attach(yolo)
            strategy.broadcast_parameters(yolo.params, yolo.moving)
            yolo._refresh_transposed()

        train_epoch_size = test_every_n_steps
        test_epoch_size = test_reader.get_image_count() / batch_size
        test_loss = list()
        names = ['loss', 'loss_xy', 'loss_wh', 'loss_obj', 'loss_class']
        train_metrics = [model.Mean('train_' + n) for n in names]
        test_metrics = [model.Mean('test_' + n) for n in names]

        current_time = datetime.datetime.now().strftime("%Y%m%dT%H%M%S")
        log_dir = os.path.join(output_folder, 'scalars-' + current_time)
        if rank == 0:
            os.makedirs(log_dir, exist_ok=True)
            for split in ('train', 'test'):
                with open(os.path.join(log_dir, split + '.csv'), 'w') as fh:
                    fh.write('step,' + ','.join(names) + '\n')

        def log_scalars(split, step, metrics):
            if rank == 0:
                with open(os.path.join(log_dir, split + '.csv'), 'a') as fh:
                    fh.write('{},{}\n'.format(step, ','.join(repr(m.result()) for m in metrics)))

        epoch = 0
        print('Running Network')
        while True:  # loop until early stopping
            print('---- Epoch: {} ----'.format(epoch))
            if epoch == 0:
                cur_train_epoch_size = min(1000, train_epoch_size)
                print('Performing Adam Optimizer learning rate warmup for {} steps'.format(cur_train_epoch_size))
                yolo.set_learning_rate(learning_rate / 10)
            else:
                cur_train_epoch_size = train_epoch_size
                yolo.set_learning_rate(learning_rate)

            start_time = time.time()
            for step, (batch_images, l1, l2, l3) in enumerate(train_dataset):
                if step > cur_train_epoch_size:
                    break
                inputs = (batch_images, (l1, l2, l3), *train_metrics)
                loss_value = yolo.dist_train_step(strategy, inputs)
                abort_on_nan(loss_value, 'Training Loss went to NaN, try a lower learning rate')
                print('Train Epoch {}: Batch {}/{}: Loss {}'.format(epoch, step, train_epoch_size, train_metrics[0].result()))
                log_scalars('train', int(epoch * train_epoch_size + step), train_metrics)
                for m in train_metrics:
                    m.reset_states()

            epoch_test_loss = list()
            for step, (batch_images, l1, l2, l3) in enumerate(test_dataset):
                if step > test_epoch_size:
                    break
                inputs = (batch_images, (l1, l2, l3), *test_metrics)
                loss_value = yolo.dist_test_step(strategy, inputs)
                abort_on_nan(loss_value, 'Test Loss went to NaN')
                epoch_test_loss.append(float(loss_value))
            test_loss.append(np.mean(epoch_test_loss))
            print('Test Epoch: {}: Loss = {}'.format(epoch, test_metrics[0].result()))
            log_scalars('test', int((epoch + 1) * train_epoch_size), test_metrics)
            for m in test_metrics:
                m.reset_states()

            if rank == 0:
                with open(os.path.join(output_folder, 'test_loss.csv'), 'w') as csvfile:
                    for v in test_loss:
                        csvfile.write(str(v))
                        csvfile.write('\n')
            print('Epoch took: {} s'.format(time.time() - start_time))

            if is_new_minimum(test_loss):
                print('Test loss improved: {}, saving checkpoint'.format(np.min(test_loss)))
                # BN moving statistics are sync-on-read: the checkpoint stores their MEAN over the replicas (App. C4); every
                # replica keeps its own running values (a collective: all ranks take part, rank 0 writes)
                saved_moving = strategy.mean_moving_stats(yolo.moving) if strategy is not None else None
                training_checkpoint_filepath = os.path.join(output_folder, 'checkpoint', 'ckpt.npz')
                if rank == 0:
                    os.makedirs(os.path.dirname(training_checkpoint_filepath), exist_ok=True)
                    yolo.save_weights(training_checkpoint_filepath, moving=saved_moving)

            print('Best Current Epoch Selection:')
            print('Test Loss:')
            print(test_loss)
            print('Best epoch: {}'.format(best_epoch_of(test_loss)))
            if should_stop(test_loss, early_stopping_count):
                break
            epoch = epoch + 1
            if max_epochs is not None and epoch >= max_epochs:
                break
    finally:
        print('Shutting down train_reader')
        train_reader.shutdown()
        print('Shutting down test_reader')
        test_reader.shutdown()

    if training_checkpoint_filepath is not None and rank == 0:
        print('Converting checkpoint into Saved_Model')
        from yolo3 import model
        best = model.YoloV3(global_batch_size, train_reader.get_image_size(), number_classes, anchors, learning_rate)
        best.load_weights(training_checkpoint_filepath)
        os.makedirs(os.path.join(output_folder, 'saved_model'), exist_ok=True)
        best.save_weights(os.path.join(output_folder, 'saved_model', 'yolov3.npz'))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    parser = argparse.ArgumentParser(prog='train_yolo', description='Script which trains a yolo_v3 model')
    parser.add_argument('--batch_size', dest='batch_size', type=int, help='training batch size', default=8)
    parser.add_argument('--learning_rate', dest='learning_rate', type=float, default=1e-4)
    parser.add_argument('--test_every_n_steps', dest='test_every_n_steps', type=int, help='number of gradient update steps to take between test epochs', default=1000)
    parser.add_argument('--train_database', dest='train_database_filepath', type=str, help='lmdb database to use for (Required)', required=True)
    parser.add_argument('--test_database', dest='test_database_filepath', type=str, help='lmdb database to use for testing (Required)', required=True)
    parser.add_argument('--output_dir', dest='output_folder', type=str, help='Folder where outputs will be saved (Required)', required=True)
    parser.add_argument('--early_stopping', dest='terminate_after_num_epochs_without_test_loss_improvement', type=int, default=10)
    parser.add_argument('--use_augmentation', dest='use_augmentation', type=int, default=1)
    parser.add_argument('--reader_count', dest='reader_count', type=int, default=None, help='(addition) reader processes per GPU; default: 3 (as the reference) up to 12 when the host has the cores')
    parser.add_argument('--max_epochs', dest='max_epochs', type=int, default=None, help='(addition) stop after this many epochs')
    parser.add_argument('--backend', dest='backend', type=str, default='nccl', help='(addition) torch.distributed backend under torch.distributed.run: nccl (= RCCL, one GPU per rank) or gloo (rehearsal; ranks may share a GPU)')
    a = parser.parse_args()
    print('Arguments:')
    for k, v in vars(a).items():
        print('{} = {}'.format(k, v))
    train_model(a.batch_size, a.test_every_n_steps, a.train_database_filepath, a.test_database_filepath, a.output_folder,
                a.terminate_after_num_epochs_without_test_loss_improvement, a.learning_rate, bool(a.use_augmentation), a.max_epochs, a.reader_count, a.backend)
