# Entry points of this repo.  The variable NAMES are the reference's (its `make train ...` / `make inference ...` command lines keep
# working: MODEL, DEPTHMUL, CKPT, IMG, BATCH, ...); everything else is this build's.
#
#   make build       hipcc --offload-arch=gfx950 -> k210_yolo_framework_amd/csrc/libyolo_hip.so (+ the CPU oracle used by the tests)
#   make test        CPU suite;  `python -m pytest tests -m gpu` needs an MI355X
#   make bench       images/sec, yolo_mobilev1-0.75, 32 frames per step (GPUS=N runs one rank per GPU through torchrun)
#   make inference   MODEL=... DEPTHMUL=... CKPT=weights.h5|.npz IMG=picture.jpg
#   make train       MODEL=... DEPTHMUL=... BATCH=16 MAXEP=10 [SYNTHETIC=256]
#   make anchors     DATASET=voc ANCNUM=3 [LOW='0.0 0.0' HIGH='1.0 1.0']   (reference Makefile:78-87: k-means anchors from data/<set>_img_ann.npy)

PY            ?= python3
MODEL         ?= yolo_mobilev1
DEPTHMUL      ?= 0.75
CLSNUM        ?= 20
DATASET       ?= voc
IMGSIZE       ?= 224 320
OUTSIZE       ?= 7 10 14 20
OBJTHRESH     ?= 0.7
IOUTHRESH     ?= 0.5
CKPT          ?= ""
IMG           ?= data/synthetic_320x224.jpg
# training only
BATCH         ?= 32
MAXEP         ?= 10
ILR           ?= 0.0005
LRDECAYFACTOR ?= 0
OBJWEIGHT     ?= 1
NOOBJWEIGHT   ?= 1
WHWEIGHT      ?= 1
SPLITFACTOR   ?= 0.05
IAA           ?= False
PRUNE         ?= False
SYNTHETIC     ?= 0
GPUS          ?= 1
# anchors only (reference Makefile:27-29)
ANCNUM        ?= 3
LOW           ?= 0.0 0.0
HIGH          ?= 1.0 1.0

NET_ARGS   = --train_set $(DATASET) --class_num $(CLSNUM) --model_def $(MODEL) --depth_multiplier $(DEPTHMUL) \
             --image_size $(IMGSIZE) --output_size $(OUTSIZE) --obj_thresh $(OBJTHRESH) --iou_thresh $(IOUTHRESH)
TRAIN_ARGS = --pre_ckpt $(CKPT) --augmenter $(IAA) --batch_size $(BATCH) --rand_seed 3 --max_nrof_epochs $(MAXEP) \
             --init_learning_rate $(ILR) --learning_rate_decay_factor $(LRDECAYFACTOR) --obj_weight $(OBJWEIGHT) \
             --noobj_weight $(NOOBJWEIGHT) --wh_weight $(WHWEIGHT) --vaildation_split $(SPLITFACTOR) --log_dir log \
             --is_prune $(PRUNE) --synthetic $(SYNTHETIC)
ifeq ($(GPUS),1)
LAUNCH = $(PY)
else
LAUNCH = $(PY) -m torch.distributed.run --nnodes=1 --nproc-per-node $(GPUS) --master-addr 127.0.0.1 --master-port 29533
endif

.PHONY: all build test bench inference train anchors
all:
	@echo 'targets: build | test | bench | inference | train   (see the header of this Makefile)'

build:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:
	$(PY) -m pytest tests -x -q -m "not gpu"

bench:
	$(LAUNCH) bench.py --gpus $(GPUS)

inference:
	$(PY) keras_inference.py $(CKPT) $(IMG) $(NET_ARGS)

train:
	$(LAUNCH) keras_train.py $(NET_ARGS) $(TRAIN_ARGS)

# reference Makefile:78-87 (same flags; --is_random True as there)
anchors:
	$(PY) ./make_anchor_list.py $(DATASET) --max_iters 10 --is_random True --in_hw $(IMGSIZE) --out_hw $(OUTSIZE) --anchor_num $(ANCNUM) \
		--low $(LOW) --high $(HIGH)
