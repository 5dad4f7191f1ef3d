# Same variables and targets as the reference Makefile (reference Makefile:1-93) for the hot path.
CKPT=""
IAA=False
ILR=0.0005
CLSNUM=20
BATCH=32
DATASET=voc
MAXEP=10
MODEL=yolo_mobilev1
DEPTHMUL=0.75
LRDECAYFACTOR=0
OBJWEIGHT=1
NOOBJWEIGHT=1
WHWEIGHT=1
IMG=data/people.jpg
SPLITFACTOR=0.05
OBJTHRESH=0.7
IOUTHRESH=0.5
IMGSIZE=224 320
OUTSIZE=7 10 14 20
GPUS=1
PRUNE=False
SYNTHETIC=0

all:
	@echo please use \"make build\", \"make inference\", \"make bench\", \"make test\" ...

build:
	python3 -c "import __graft_entry__ as g; g.build()"

inference:
	python3 ./keras_inference.py \
			${CKPT} \
			${IMG} \
			--train_set ${DATASET} \
			--class_num ${CLSNUM} \
			--model_def ${MODEL} \
			--depth_multiplier ${DEPTHMUL} \
			--obj_thresh ${OBJTHRESH} \
			--iou_thresh ${IOUTHRESH} \
			--image_size ${IMGSIZE} \
			--output_size ${OUTSIZE}

train:
	python3 ./keras_train.py \
			--train_set ${DATASET} \
			--class_num ${CLSNUM} \
			--pre_ckpt ${CKPT} \
			--model_def ${MODEL} \
			--depth_multiplier ${DEPTHMUL} \
			--augmenter ${IAA} \
			--image_size ${IMGSIZE} \
			--output_size ${OUTSIZE} \
			--batch_size ${BATCH} \
			--rand_seed 3 \
			--max_nrof_epochs ${MAXEP} \
			--init_learning_rate ${ILR} \
			--learning_rate_decay_factor ${LRDECAYFACTOR} \
			--obj_weight ${OBJWEIGHT} \
			--noobj_weight ${NOOBJWEIGHT} \
			--wh_weight ${WHWEIGHT} \
			--obj_thresh ${OBJTHRESH} \
			--iou_thresh ${IOUTHRESH} \
			--vaildation_split ${SPLITFACTOR} \
			--log_dir log \
			--is_prune ${PRUNE} \
			--synthetic ${SYNTHETIC}

bench:
	python3 bench.py --gpus ${GPUS}

test:
	python3 -m pytest tests -x -q -m "not gpu"

.PHONY: all build inference train bench test
