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
	@echo "make train (keras_train.py: YOLO loss + Adam + RCCL gradient all-reduce) is SURVEY.md 8(a) rows T1-T5 /"
	@echo "BASELINE config 4; the backward kernels are not built yet in this round - see DESIGN.md 'what comes next'."
	@false

bench:
	python3 bench.py --gpus ${GPUS}

test:
	python3 -m pytest tests -x -q -m "not gpu"

.PHONY: all build inference train bench test
