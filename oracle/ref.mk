# oracle/ref.mk — builds oracle/_ref/* from reference sources where they lie. See oracle/Makefile.
all:
	@true
