/* stand-in for <GL/freeglut.h>: the names reference standalone.c:62-108 uses (its GLUT window mode) */
#pragma once
void glutSwapBuffers(void);
void glutExit(void);
void glutPostRedisplay(void);
void glutDisplayFunc(void (*callback)(void));
void glutKeyboardFunc(void (*callback)(unsigned char key, int x, int y));
void glutReshapeFunc(void (*callback)(int width, int height));
void glutMainLoop(void);
